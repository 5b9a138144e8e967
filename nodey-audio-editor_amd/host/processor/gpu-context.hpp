// processor/gpu-context.hpp — how a processor fiber talks to the C ABI without ever blocking the OS thread
// (all fibers share one thread: /root/reference/src/infra/runner.cpp:65-69).
//
// Every RUNNING NODE owns a context (gpu::Node, the first local of its process_payload): its own HIP stream, so
//   * a node waits for ITS OWN batch — gpu::wait polls the node's stream, not a stream shared with every other node of the graph
//     (round 3: one process-wide context; a node whose batch had finished kept yielding while any other node's was queued);
//   * batches of different nodes overlap on the device;
//   * nodes MAY be spread over the GPUs of the machine: with $NAE_DEVICES=n node k of a run goes to device k mod min(n, nae_device_count())
//     (the reference runs the whole graph in ONE process on ONE thread, src/infra/runner.cpp:142-154; every hop between nodes passes
//     host frames anyway — Audio_stream carries AVFrames — so any node can sit on any device).  WITHOUT $NAE_DEVICES every node runs on
//     ONE device ($NAE_DEVICE, default 0): more than one device from one process has never run on hardware here (one-GPU leases; the
//     two-device test skips), so spreading is opt-in until it has.
// Threads: one Runner thread drives all nodes of a run (cooperative fibers), but the editor builds the next Runner before it destroys the
// previous one (src/frontend/app.cpp:2028; each Runner starts its fibers on a fresh detached std::thread, src/infra/runner.cpp:153), so a
// finishing run's ~Node can meet a starting run's Node(): the pool, the counters and the device round-robin are guarded by ONE mutex.
// A context itself is still driven by one thread at a time (include/nae_gpu.h, Threads).
// The node's context is found through one pointer of fiber-local storage (boost::fibers::fiber_specific_ptr in the reference's
// scheduler), so helpers deep inside a processor need no extra argument.
#pragma once
#include "../../../include/nae_gpu.h"
#include "../infra/fiber.hpp"
#include "../infra/processor.hpp"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace processor::gpu
{
	// what the plugin-boundary benchmark reads (tests/host/selftest.cpp `bench`): waits and polls of all nodes, and how many
	// nodes had a batch in flight at the same moment
	struct Flight_stats
	{
		size_t waits = 0, polls = 0, nodes = 0, contexts_created = 0;
		int in_flight = 0, max_in_flight = 0;
		int devices_used = 0;
		double setup_seconds = 0.0;  // spent creating contexts and (re)allocating device / page-locked buffers
	};
	struct Setup_timer
	{
		std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
		~Setup_timer();
	};
	inline Flight_stats& flight_stats()
	{
		static Flight_stats s;
		return s;
	}
	// guards context_pool(), pick_device()'s counter and the creation / return of contexts (see "Threads" above); the per-wait counters of
	// Flight_stats are statistics of ONE runner thread (the benchmark's) and stay unguarded
	inline std::mutex& shared_state_mutex()
	{
		static std::mutex m;
		return m;
	}

	inline Setup_timer::~Setup_timer() { flight_stats().setup_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }

	inline int pick_device()
	{
		static unsigned next = 0;  // (under shared_state_mutex)
		const char* dev = std::getenv("NAE_DEVICE");
		const char* lim = std::getenv("NAE_DEVICES");
		// one device unless the integrator asks for more: N > 1 devices per process are unmeasured on hardware (include/nae_gpu.h, Devices)
		if (lim == nullptr || std::atoi(lim) <= 1) return dev ? std::atoi(dev) : 0;
		int n = nae_device_count();
		if (std::atoi(lim) < n) n = std::atoi(lim);
		if (n <= 0) return 0;  // nae_ctx_create reports the missing device
		std::lock_guard<std::mutex> lock(shared_state_mutex());
		const int d = (int)(next++ % (unsigned)n);
		if (d + 1 > flight_stats().devices_used) flight_stats().devices_used = d + 1;
		return d;
	}

	// contexts that finished nodes have given back, per device: the editor starts a new Runner for every preview / export
	// (frontend/app.cpp:2001-2094), and a context (stream + tables) costs a millisecond or two to create
	inline std::map<int, std::vector<nae_ctx*>>& context_pool()
	{
		static std::map<int, std::vector<nae_ctx*>> pool;
		return pool;
	}

	inline nae_ctx* create_context(int device)
	{
		{
			std::lock_guard<std::mutex> lock(shared_state_mutex());
			auto& free_list = context_pool()[device];
			if (!free_list.empty())
			{
				nae_ctx* ctx = free_list.back();
				free_list.pop_back();
				return ctx;
			}
			flight_stats().contexts_created++;
			if (flight_stats().devices_used < 1) flight_stats().devices_used = 1;
		}
		Setup_timer timer;
		nae_ctx* ctx = nullptr;
		const int rc = nae_ctx_create(device, &ctx);
		if (rc != NAE_OK)
			throw infra::Processor::Runtime_error(
				"GPU context creation failed",
				"The MI355X processors need a HIP device; there is no CPU fallback.",
				infra::fmt("nae_ctx_create(%d) returned %d", device, rc)
			);
		return ctx;
	}

	// a finished node's context goes back to the pool — unless its last synchronisation failed: a context in error state is destroyed,
	// so the next run never inherits it
	inline void return_context(int device, nae_ctx* ctx)
	{
		if (ctx == nullptr) return;
		if (nae_sync(ctx) != NAE_OK)
		{
			nae_ctx_destroy(ctx);
			return;
		}
		std::lock_guard<std::mutex> lock(shared_state_mutex());
		context_pool()[device].push_back(ctx);  // kept for the next run's nodes (destroyed with the process)
	}

	// the context of one running node; declare it FIRST in process_payload (device buffers and handles die before it)
	class Node
	{
		nae_ctx* ctx_;
		void* outer;

	  public:

		const int device;
		bool pending = false;  // a batch of this node is being waited for

		Node() : ctx_(nullptr), outer(nae_fiber::this_fiber::local()), device(pick_device())
		{
			ctx_ = create_context(device);
			nae_fiber::this_fiber::local() = this;
			std::lock_guard<std::mutex> lock(shared_state_mutex());
			flight_stats().nodes++;
		}
		Node(const Node&) = delete;
		Node& operator=(const Node&) = delete;
		~Node()
		{
			nae_fiber::this_fiber::local() = outer;
			return_context(device, ctx_);
		}
		nae_ctx* ctx() const { return ctx_; }
	};

	inline Node* current_node() { return static_cast<Node*>(nae_fiber::this_fiber::local()); }

	// the running node's context; outside a node (unit tests that call a helper directly): one context per process on
	// device $NAE_DEVICE (default 0)
	inline nae_ctx* context()
	{
		if (Node* node = current_node()) return node->ctx();
		static std::once_flag once;
		static nae_ctx* fallback = nullptr;
		std::call_once(once, [] {
			const char* dev = std::getenv("NAE_DEVICE");
			fallback = create_context(dev ? std::atoi(dev) : 0);
		});
		return fallback;
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

	// poll-and-yield until everything THIS NODE has queued is done (or stop is requested)
	inline void wait(const std::atomic<bool>& stop_token)
	{
		nae_ctx* ctx = context();
		Node* node = current_node();
		Flight_stats& fs = flight_stats();
		fs.waits++;
		for (;;)
		{
			const int r = nae_poll(ctx);
			fs.polls++;
			if (r != 0 || stop_token)
			{
				if (node && node->pending) { node->pending = false; fs.in_flight--; }
				if (r < 0) check(r, "nae_poll");
				if (r == 0) nae_sync(ctx);
				return;
			}
			if (node && !node->pending)
			{
				node->pending = true;
				if (++fs.in_flight > fs.max_in_flight) fs.max_in_flight = fs.in_flight;
			}
			nae_fiber::this_fiber::yield();
		}
	}

	// grow-only device scratch of the node that created it
	class Device_buffer
	{
		nae_ctx* ctx;
		void* ptr = nullptr;
		size_t bytes = 0;

	  public:

		Device_buffer() : ctx(context()) {}
		Device_buffer(const Device_buffer&) = delete;
		Device_buffer& operator=(const Device_buffer&) = delete;
		~Device_buffer() { if (ptr) { nae_sync(ctx); nae_free(ctx, ptr); } }
		void* reserve(size_t want)
		{
			if (want > bytes)
			{
				Setup_timer timer;
				if (ptr) { nae_sync(ctx); nae_free(ctx, ptr); ptr = nullptr; }
				check(nae_malloc(ctx, want + want / 2 + 256, &ptr), "nae_malloc");
				bytes = want + want / 2 + 256;
			}
			return ptr;
		}
		template <typename T> T* as() { return static_cast<T*>(ptr); }
	};

	// grow-only PAGE-LOCKED host staging of the node that created it.  Audio_frame buffers are pageable (av_frame_get_buffer /
	// malloc): a hipMemcpyAsync from or to them is staged by the runtime and a download BLOCKS the calling thread until the stream
	// has reached it — on the one thread all fibers share.  So a node copies the frames of a batch into / out of one pinned buffer
	// on the CPU (tens of KB: microseconds) and moves the batch with ONE asynchronous copy each way; the fiber then really yields
	// while the GPU works, and other nodes' batches overlap with it.
	class Pinned_buffer
	{
		nae_ctx* ctx;
		void* ptr = nullptr;
		size_t bytes = 0;

	  public:

		Pinned_buffer() : ctx(context()) {}
		Pinned_buffer(const Pinned_buffer&) = delete;
		Pinned_buffer& operator=(const Pinned_buffer&) = delete;
		~Pinned_buffer() { if (ptr) { nae_sync(ctx); nae_free_host(ctx, ptr); } }
		void* reserve(size_t want)
		{
			if (want > bytes)
			{
				Setup_timer timer;
				if (ptr) { nae_sync(ctx); nae_free_host(ctx, ptr); ptr = nullptr; }
				check(nae_malloc_host(ctx, want + want / 2 + 4096, &ptr), "nae_malloc_host");
				bytes = want + want / 2 + 4096;
			}
			return ptr;
		}
	};
}
