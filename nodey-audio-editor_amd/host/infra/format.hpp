// infra/format.hpp — printf-style stand-in for std::format (libstdc++-11 has no <format>), and the reference's
// THROW_LOGIC_ERROR (include/utility/logic-error-utility.hpp:2-12) on top of it.
#pragma once
#include <cstdio>
#include <stdexcept>
#include <string>

namespace infra
{
	inline std::string fmt(const char* text) { return text; }

	template <typename... Args>
	std::string fmt(const char* pattern, Args... args)
	{
		char buffer[1024];
		std::snprintf(buffer, sizeof buffer, pattern, args...);
		return buffer;
	}

	[[noreturn]] inline void throw_logic_error(const char* file, int line, const std::string& what)
	{
		throw std::logic_error(std::string(file) + "(" + std::to_string(line) + "): " + what);
	}
}

#define THROW_LOGIC_ERROR(...) ::infra::throw_logic_error(__FILE__, __LINE__, ::infra::fmt(__VA_ARGS__))
