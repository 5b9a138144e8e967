#include "fiber.hpp"

#include <ucontext.h>

#include <cstdlib>
#include <stdexcept>

namespace nae_fiber
{
	struct Fiber
	{
		ucontext_t ctx{};
		std::function<void()> body;
		std::unique_ptr<char[]> stack;
		bool done = false;
		void* local = nullptr;
	};

	struct Scheduler::Impl
	{
		ucontext_t main_ctx{};
		std::vector<std::unique_ptr<Fiber>> fibers;
		Fiber* current = nullptr;
		size_t switches = 0;
	};

	static thread_local Scheduler::Impl* active = nullptr;

	static void trampoline()
	{
		Fiber* f = active->current;
		f->body();
		f->done = true;
		swapcontext(&f->ctx, &active->main_ctx);
	}

	Scheduler::Scheduler() : impl(std::make_unique<Impl>()) {}
	Scheduler::~Scheduler() = default;

	void Scheduler::spawn(std::function<void()> body, size_t stack_bytes)
	{
		auto f = std::make_unique<Fiber>();
		f->body = std::move(body);
		f->stack = std::make_unique<char[]>(stack_bytes);
		if (getcontext(&f->ctx) != 0) throw std::runtime_error("getcontext failed");
		f->ctx.uc_stack.ss_sp = f->stack.get();
		f->ctx.uc_stack.ss_size = stack_bytes;
		f->ctx.uc_link = nullptr;
		makecontext(&f->ctx, trampoline, 0);
		impl->fibers.push_back(std::move(f));
	}

	void Scheduler::run()
	{
		Impl* previous = active;
		active = impl.get();
		bool any = true;
		while (any)
		{
			any = false;
			for (auto& f : impl->fibers)
			{
				if (f->done) continue;
				any = true;
				impl->current = f.get();
				++impl->switches;
				swapcontext(&impl->main_ctx, &f->ctx);
			}
		}
		impl->current = nullptr;
		active = previous;
	}

	size_t Scheduler::switches() const { return impl->switches; }

	void*& this_fiber::local()
	{
		static thread_local void* thread_slot = nullptr;
		if (active == nullptr || active->current == nullptr) return thread_slot;
		return active->current->local;
	}

	void this_fiber::yield()
	{
		if (active == nullptr || active->current == nullptr) return;
		Fiber* f = active->current;
		swapcontext(&f->ctx, &active->main_ctx);
	}
}
