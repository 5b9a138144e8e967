// processor/gpu-context.hpp — how a processor fiber talks to the C ABI without ever blocking the OS thread
// (all fibers share one thread: /root/reference/src/infra/runner.cpp:65-69).
#pragma once
#include "../../../include/nae_gpu.h"
#include "../infra/fiber.hpp"
#include "../infra/processor.hpp"

#include <atomic>
#include <cstdlib>

namespace processor::gpu
{
	// process-wide context on device $NAE_DEVICE (default 0); Runtime_error if there is no usable GPU —
	// the adapter has no CPU fallback
	inline nae_ctx* context()
	{
		static nae_ctx* ctx = nullptr;
		if (ctx == nullptr)
		{
			const char* dev = std::getenv("NAE_DEVICE");
			const int rc = nae_ctx_create(dev ? std::atoi(dev) : 0, &ctx);
			if (rc != NAE_OK)
				throw infra::Processor::Runtime_error(
					"GPU context creation failed",
					"The MI355X processors need a HIP device; there is no CPU fallback.",
					infra::fmt("nae_ctx_create returned %d", rc)
				);
		}
		return ctx;
	}

	// non-zero status -> the reference's user-facing error type (include/infra/processor.hpp:64-77)
	inline void check(int rc, const char* what)
	{
		if (rc == NAE_OK) return;
		throw infra::Processor::Runtime_error(
			infra::fmt("GPU call failed: %s", what),
			"The GPU audio kernel library reported an error.",
			infra::fmt("status %d: %s", rc, nae_last_error(context()))
		);
	}

	// poll-and-yield until the context's stream is idle (or stop is requested)
	inline void wait(const std::atomic<bool>& stop_token)
	{
		for (;;)
		{
			const int r = nae_poll(context());
			if (r == 1) return;
			if (r < 0) check(r, "nae_poll");
			if (stop_token) { nae_sync(context()); return; }
			nae_fiber::this_fiber::yield();
		}
	}

	// grow-only device scratch
	class Device_buffer
	{
		void* ptr = nullptr;
		size_t bytes = 0;

	  public:

		Device_buffer() = default;
		Device_buffer(const Device_buffer&) = delete;
		Device_buffer& operator=(const Device_buffer&) = delete;
		~Device_buffer() { if (ptr) nae_free(context(), ptr); }
		void* reserve(size_t want)
		{
			if (want > bytes)
			{
				if (ptr) { nae_sync(context()); nae_free(context(), ptr); ptr = nullptr; }
				check(nae_malloc(context(), want + want / 2 + 256, &ptr), "nae_malloc");
				bytes = want + want / 2 + 256;
			}
			return ptr;
		}
		template <typename T> T* as() { return static_cast<T*>(ptr); }
	};
}
