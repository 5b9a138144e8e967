// bimix-align.hpp — how Audio_bimix_v2 lines the left and the right input up on their presentation times.
//
// The reference does this inside its processing loop (/root/reference/src/processor/audio-bimix.cpp:777-872); the rule is
// kept — saved projects must produce the same frames — but stated as a pure function of the two pending spans, so that it can
// be tested without a runner (tests/host/selftest.cpp, `selftest cpu`):
//
//   * the span that begins first (the right one on a tie) is `first`;
//   * if it ends before the other begins, it is emitted alone and used up;
//   * otherwise the output frame is `solo` samples of `first` alone (the other channel silent) followed by `both` samples of
//     the two together, where solo = round((begin_other - begin_first) * rate) and both = round((earlier end - begin_other)
//     * rate), limited by what the two spans hold; the span that ends first is used up (whatever rounding left of it is
//     dropped, as the reference's pop_front does), the other one loses the samples that were played.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>

namespace processor::bimix
{
	struct Span
	{
		double begin;       // presentation time of the first pending sample, seconds
		std::size_t count;  // pending samples
		double end(int rate) const { return begin + double(count) / rate; }
	};

	struct Step
	{
		int first;            // 0: the left span starts the output frame, 1: the right one
		std::size_t solo;     // samples of `first` alone
		std::size_t both;     // samples of both sides
		bool used_up[2];      // left / right: the span is finished (remove it)
		std::size_t played[2];// left / right: samples consumed from a span that is NOT used up
	};

	inline Step align_step(const Span& left, const Span& right, int rate)
	{
		Step s{};
		s.first = left.begin < right.begin ? 0 : 1;
		const Span& a = s.first == 0 ? left : right;   // starts first
		const Span& b = s.first == 0 ? right : left;
		const int ia = s.first, ib = 1 - s.first;
		if (a.end(rate) <= b.begin)
		{
			s.solo = a.count;
			s.used_up[ia] = true;
			return s;
		}
		const double stop = std::min(a.end(rate), b.end(rate));
		s.solo = static_cast<std::size_t>(std::round((b.begin - a.begin) * rate));
		s.both = static_cast<std::size_t>(std::round((stop - b.begin) * rate));
		s.both = std::min({s.both, a.count - s.solo, b.count});
		if (a.end(rate) <= b.end(rate))
		{
			s.used_up[ia] = true;
			s.played[ib] = s.both;
		}
		else
		{
			s.used_up[ib] = true;
			s.played[ia] = s.solo + s.both;
		}
		return s;
	}
}
