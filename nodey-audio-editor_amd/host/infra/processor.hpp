// infra/processor.hpp — host-side mirror of the reference's plugin interface
// (/root/reference/include/infra/processor.hpp:26-176): same class, member and function names, same argument
// meaning, same error behaviour, so a processor written against the reference compiles against this header and
// the GPU processors in ../processor/ drop into src/register.cpp unchanged.  Differences, all forced by the
// toolchain of this image (g++ 11, no JsonCpp): std::format -> snprintf helper, Json::Value -> json_mini.hpp.
#pragma once

#include <any>
#include <atomic>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <typeinfo>
#include <vector>

#include "json_mini.hpp"

namespace infra
{
	using Id_t = int;

	inline std::string fmt(const char* f) { return f; }
	template <typename... Args>
	std::string fmt(const char* f, Args... args)
	{
		char buf[1024];
		std::snprintf(buf, sizeof buf, f, args...);
		return buf;
	}

// reference: include/utility/logic-error-utility.hpp:2-12
#define THROW_LOGIC_ERROR(...) \
	{ throw std::logic_error(std::string(__FILE__) + "(" + std::to_string(__LINE__) + "): " + ::infra::fmt(__VA_ARGS__)); }

	class Processor
	{
	  public:

		// reference :32-39
		class Product
		{
		  public:

			Product() = default;
			virtual ~Product() = default;
			const std::type_info& get_typeinfo() const { return typeid(*this); }
		};

		// reference :42-49
		struct Pin_attribute
		{
			std::string identifier;
			std::string display_name;
			std::reference_wrapper<const std::type_info> type;
			bool is_input;
			std::function<std::shared_ptr<Product>()> generate_func;
		};

		// reference :52-59
		struct Info
		{
			std::string identifier;
			std::string display_name;
			bool singleton = false;
			std::function<std::unique_ptr<Processor>()> generate;
			std::string description;
		};

		// reference :64-77 — message layout "{} (Detail: {}) (Explanation: {})"
		struct Runtime_error : public std::runtime_error
		{
			std::string message, explanation, detail;

			Runtime_error(std::string message, std::string explanation, std::string detail = "") :
				std::runtime_error(message + " (Detail: " + detail + ") (Explanation: " + explanation + ")"),
				message(std::move(message)),
				explanation(std::move(explanation)),
				detail(std::move(detail))
			{
			}
		};

		static std::map<std::string, Processor::Info> processor_map;  // reference :80

		Processor() = default;
		virtual ~Processor() = default;

		virtual std::vector<Processor::Pin_attribute> get_pin_attributes() const = 0;  // :86
		virtual Processor::Info get_processor_info_non_static() const = 0;             // :90
		virtual Json::Value serialize() const = 0;                                     // :93
		virtual void deserialize(const Json::Value& value) = 0;                        // :96
		virtual void draw_title() {}                                                   // :99  (GUI: no-op here)
		virtual bool draw_content(bool readonly) { (void)readonly; return false; }      // :104 (GUI: no-op here)

		// reference :108-113
		virtual void process_payload(
			const std::map<std::string, std::shared_ptr<Processor::Product>>& input,
			const std::map<std::string, std::set<std::shared_ptr<Processor::Product>>>& output,
			const std::atomic<bool>& stop_token,
			std::any& user_data
		) = 0;

		// reference :116-129: duplicate identifier => logic_error
		template <typename T>
		static void register_processor()
		{
			const Info processor_info = T::get_processor_info();
			if (processor_map.count(processor_info.identifier))
				THROW_LOGIC_ERROR("Processor with identifier '%s' already registered", processor_info.identifier.c_str());
			processor_map[processor_info.identifier] = T::get_processor_info();
		}
	};

	// reference :134-155
	template <typename T>
	std::optional<std::reference_wrapper<T>> get_input_item(
		const std::map<std::string, std::shared_ptr<Processor::Product>>& input,
		const std::string& key
	)
	{
		const auto find = input.find(key);
		if (find == input.end()) return std::nullopt;
		if (find->second == nullptr) THROW_LOGIC_ERROR("Found nullptr in input map for key '%s'", key.c_str());
		if (find->second->get_typeinfo() != typeid(T))
			THROW_LOGIC_ERROR(
				"Type mismatch in input map for key '%s', expected %s, got %s",
				key.c_str(),
				typeid(T).name(),
				find->second->get_typeinfo().name()
			);
		return *std::dynamic_pointer_cast<T>(find->second);
	}

	// reference :158-176
	template <typename T>
	std::set<std::shared_ptr<T>> get_output_item(
		const std::map<std::string, std::set<std::shared_ptr<Processor::Product>>>& output,
		const std::string& key
	)
	{
		const auto find = output.find(key);
		if (find == output.end()) THROW_LOGIC_ERROR("Key '%s' not found in output map", key.c_str());
		std::set<std::shared_ptr<T>> output_set;
		for (auto& item : find->second)
		{
			if (item == nullptr) THROW_LOGIC_ERROR("Found nullptr in output map for key '%s'", key.c_str());
			output_set.emplace(std::dynamic_pointer_cast<T>(item));
		}
		return output_set;
	}

	// reference :181, src/register.cpp:14-24
	void register_all_processors();
}
