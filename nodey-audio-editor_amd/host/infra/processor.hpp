// infra/processor.hpp — the plugin boundary of the editor, restated for the GPU processors.
//
// A node of the reference is a subclass of infra::Processor (/root/reference/include/infra/processor.hpp:26-130);
// everything the GPU drop-ins touch of that contract is declared here under the reference's names so that
// ../processor/*.cpp compile unchanged against the real header:
//   Product (:32-39) · Pin_attribute (:42-49) · Info (:52-59) · Runtime_error (:64-77) · processor_map (:80)
//   the pure virtuals (:86-113) · register_processor<T> (:116-129) · get_input_item / get_output_item (:134-176)
// What differs is forced by this image's toolchain: Json::Value comes from json_mini.hpp, std::format from format.hpp.
// The two GUI hooks (draw_title, draw_content) are PURE virtual here as in the reference (:99-104): every GPU processor declares them, and
// ../processor/draw-headless.cpp holds their bodies without a GUI — the one file an integrator replaces with the reference's own ImGui
// bodies, which are the only code that changes a node's parameters while the editor runs.
#pragma once

#include <any>
#include <atomic>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <string>
#include <typeinfo>
#include <vector>

#include "format.hpp"
#include "json_mini.hpp"

namespace infra
{
	using Id_t = int;

	class Processor
	{
	  public:

		// what travels along a link; nodes recover the concrete type through get_typeinfo()
		class Product
		{
		  public:

			virtual ~Product() = default;
			const std::type_info& get_typeinfo() const { return typeid(*this); }
		};

		using Product_ptr = std::shared_ptr<Product>;
		using Input_map = std::map<std::string, Product_ptr>;
		using Output_map = std::map<std::string, std::set<Product_ptr>>;

		struct Pin_attribute
		{
			std::string identifier, display_name;
			std::reference_wrapper<const std::type_info> type;
			bool is_input;
			std::function<Product_ptr()> generate_func;  // the runner makes one product per link with the PRODUCER's
		};

		struct Info
		{
			std::string identifier, display_name;
			bool singleton = false;
			std::function<std::unique_ptr<Processor>()> generate;
			std::string description;
		};

		// user-facing failure: what() reads "<message> (Detail: <detail>) (Explanation: <explanation>)"
		struct Runtime_error : std::runtime_error
		{
			std::string message, explanation, detail;

			Runtime_error(const std::string& msg, const std::string& why, const std::string& more = "") :
				std::runtime_error(msg + " (Detail: " + more + ") (Explanation: " + why + ")"),
				message(msg),
				explanation(why),
				detail(more)
			{
			}
		};

		static std::map<std::string, Info> processor_map;

		virtual ~Processor() = default;

		virtual std::vector<Pin_attribute> get_pin_attributes() const = 0;
		virtual Info get_processor_info_non_static() const = 0;
		virtual Json::Value serialize() const = 0;
		virtual void deserialize(const Json::Value& value) = 0;
		virtual void draw_title() = 0;
		virtual bool draw_content(bool readonly) = 0;
		virtual void process_payload(
			const Input_map& input,
			const Output_map& output,
			const std::atomic<bool>& stop_token,
			std::any& user_data
		) = 0;

		// one entry per identifier; registering an identifier twice is a programming error
		template <typename T>
		static void register_processor()
		{
			Info info = T::get_processor_info();
			const std::string id = info.identifier;
			if (!processor_map.emplace(id, std::move(info)).second)
				THROW_LOGIC_ERROR("Processor with identifier '%s' already registered", id.c_str());
		}
	};

	namespace detail
	{
		// a product of the map as its concrete type; a null or foreign product is a wiring bug, not a user error
		template <typename T>
		std::shared_ptr<T> typed_product(const Processor::Product_ptr& product, const std::string& key, const char* side, bool check_type)
		{
			if (!product) THROW_LOGIC_ERROR("Found nullptr in %s map for key '%s'", side, key.c_str());
			if (check_type && product->get_typeinfo() != typeid(T))
				THROW_LOGIC_ERROR("Type mismatch in %s map for key '%s', expected %s, got %s", side, key.c_str(), typeid(T).name(),
								  product->get_typeinfo().name());
			return std::dynamic_pointer_cast<T>(product);
		}
	}

	// the product on input pin `key`, or nothing when the pin is not linked
	template <typename T>
	std::optional<std::reference_wrapper<T>> get_input_item(const Processor::Input_map& input, const std::string& key)
	{
		const auto it = input.find(key);
		if (it == input.end()) return std::nullopt;
		return std::ref(*detail::typed_product<T>(it->second, key, "input", true));
	}

	// every product hanging on output pin `key` (fan-out: one per outgoing link)
	template <typename T>
	std::set<std::shared_ptr<T>> get_output_item(const Processor::Output_map& output, const std::string& key)
	{
		const auto it = output.find(key);
		if (it == output.end()) THROW_LOGIC_ERROR("Key '%s' not found in output map", key.c_str());
		std::set<std::shared_ptr<T>> products;
		for (const auto& product : it->second) products.insert(detail::typed_product<T>(product, key, "output", false));
		return products;
	}

	void register_all_processors();
}
namespace processor { enum class Stretch_algorithm; }
namespace infra
{
	// same, choosing what velocity_modifier / pitch_modifier nodes WITHOUT an "algorithm" key run (processor/audio-velocity.hpp)
	void register_all_processors(processor::Stretch_algorithm default_algorithm);
}
