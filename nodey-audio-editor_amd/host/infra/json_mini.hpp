// json_mini.hpp — the handful of Json::Value operations the processors' serialize()/deserialize() use
// (reference: JsonCpp, include/infra/processor.hpp:6,93-96).  Only what the hot-path nodes touch: objects of
// bool / int / double / string members.  Not a JSON parser: project files are the editor's business (out of scope).
#pragma once
#include <map>
#include <string>
#include <variant>
#include <vector>

namespace Json
{
	class Value
	{
		std::variant<std::monostate, bool, int, double, std::string> scalar;
		std::map<std::string, Value> members;

	  public:

		Value() = default;
		Value(bool v) : scalar(v) {}
		Value(int v) : scalar(v) {}
		Value(double v) : scalar(v) {}
		Value(float v) : scalar(static_cast<double>(v)) {}
		Value(const char* v) : scalar(std::string(v)) {}
		Value(std::string v) : scalar(std::move(v)) {}

		Value& operator[](const std::string& key) { return members[key]; }
		const Value& operator[](const std::string& key) const
		{
			static const Value null_value;
			const auto it = members.find(key);
			return it == members.end() ? null_value : it->second;
		}
		bool isMember(const std::string& key) const { return members.count(key) != 0; }
		bool isNull() const { return std::holds_alternative<std::monostate>(scalar) && members.empty(); }
		bool isBool() const { return std::holds_alternative<bool>(scalar); }
		bool isInt() const { return std::holds_alternative<int>(scalar); }
		bool isString() const { return std::holds_alternative<std::string>(scalar); }
		std::string asString() const { return isString() ? std::get<std::string>(scalar) : std::string(); }
		// JsonCpp: isDouble() is true for every numeric value
		bool isDouble() const { return std::holds_alternative<double>(scalar) || std::holds_alternative<int>(scalar); }
		bool asBool() const { return isBool() ? std::get<bool>(scalar) : asDouble() != 0.0; }
		int asInt() const { return static_cast<int>(asDouble()); }
		float asFloat() const { return static_cast<float>(asDouble()); }
		double asDouble() const
		{
			if (std::holds_alternative<double>(scalar)) return std::get<double>(scalar);
			if (std::holds_alternative<int>(scalar)) return std::get<int>(scalar);
			if (std::holds_alternative<bool>(scalar)) return std::get<bool>(scalar) ? 1.0 : 0.0;
			return 0.0;
		}
		size_t size() const { return members.size(); }
		// JsonCpp: Value::Members = std::vector<std::string>, in key order
		std::vector<std::string> getMemberNames() const
		{
			std::vector<std::string> names;
			for (const auto& kv : members) names.push_back(kv.first);
			return names;
		}
	};
}
