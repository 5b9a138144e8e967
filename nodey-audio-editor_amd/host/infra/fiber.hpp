// fiber.hpp — a minimal cooperative fiber scheduler standing in for Boost.Fiber's default round-robin scheduler
// (the reference runs every node as a boost::fibers::fiber on ONE OS thread: src/infra/runner.cpp:65-69,151).
// Only what the processors use: fiber creation, join-by-run-to-completion and this_fiber::yield().
// Built on POSIX ucontext; single OS thread; exceptions thrown inside a fiber must be caught inside it.
#pragma once
#include <functional>
#include <memory>
#include <vector>

namespace nae_fiber
{
	namespace this_fiber
	{
		void yield();  // give the other fibers a turn (no-op outside a scheduler)
		// one pointer of fiber-local storage (boost::fibers::fiber_specific_ptr in the reference's Boost.Fiber); outside a
		// scheduler: one slot per OS thread
		void*& local();
	}

	class Scheduler
	{
	  public:

		Scheduler();
		~Scheduler();
		Scheduler(const Scheduler&) = delete;
		Scheduler& operator=(const Scheduler&) = delete;

		void spawn(std::function<void()> body, size_t stack_bytes = 512 * 1024);
		void run();                    // round-robin until every fiber has returned
		size_t switches() const;       // context switches performed (diagnostics)

		struct Impl;  // public only so fiber.cpp's file-local helpers can name it

	  private:

		std::unique_ptr<Impl> impl;
	};
}
