// velocity-cadence.hpp — how the velocity / pitch node cuts what its stretcher delivers into output frames.
//
// The reference's loop (/root/reference/src/processor/audio-velocity.cpp:399-436) runs in TURNS.  One turn:
//   * pop at most ONE input frame and put it into SoundTouch (:338-406);
//   * if numSamples() > min: receive ONE chunk of min(numSamples(), max) samples -> one output frame (:416-424), where
//     min = uint32(1152 / velocity), max = uint32(3 * 1152 / velocity);
//   * else, at end of input: flush() and receive EVERYTHING that is left as ONE frame, whatever its size (:425-436);
//   * yield.
// So the reference's chunk sequence is a function of the order in which puts and receive opportunities interleave — i.e. of
// how the fibers happened to be scheduled — and of numSamples() after each put.  What every schedule shares: a steady-state
// chunk n satisfies min < n <= max, the chunks concatenate to the stretcher's output, and frame k's pts is the float-typed
// microsecond clock of the samples in front of it (:238,249,306-311).
//
// The mirror (audio-velocity.cpp, this directory) puts ALL frames that are already waiting as one block (a 9-KB frame per GPU
// launch and wait would be all overhead) and then applies the reference's receive rule repeatedly until it no longer fires:
// `drain` below.  When frames arrive one at a time this IS the reference's turn sequence as long as one put never makes more
// than min + max samples available (checked for the tested settings in tests/host/selftest.cpp: `selftest cpu`, cadence case);
// under batching the chunk BOUNDARIES differ from a one-frame-per-turn run — larger chunks, up to max, earlier — while sizes
// stay inside (min, max] and samples and their order are identical.  The flush remainder is cut into chunks of at most max
// instead of one arbitrarily long frame (a deviation: the reference's last frame can exceed max).
// Downstream the mixer cuts its rounds by the shortest front frame (audio-amix.cpp:192-195) and leaves the rest of a longer
// frame inside its resampler's FIFO (swr_convert with a smaller output capacity, :263-269): other chunk boundaries give other
// round boundaries, never other samples.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace processor::cadence
{
	struct Bounds
	{
		uint32_t min_samples, max_samples;
	};

	// audio-velocity.cpp:416-417 (time_ratio is a double holding 1.0f / velocity, :291; the products are truncated to uint32_t)
	inline Bounds bounds(float velocity)
	{
		const double time_ratio = 1.0f / velocity;
		return {static_cast<uint32_t>(time_ratio * 1152), static_cast<uint32_t>(time_ratio * 1152 * 3)};
	}

	// the reference's receive rule of ONE turn (:419-423): the chunk it takes with `available` samples queued, 0 = none
	inline std::size_t reference_receive(std::size_t available, const Bounds& b)
	{
		return available > b.min_samples ? std::min<std::size_t>(available, b.max_samples) : 0;
	}

	// the mirror's turn: the reference's rule applied until it no longer fires (floor = min_samples), or — behind flush(), with
	// floor = 0 — until nothing is left
	inline std::vector<std::size_t> drain(std::size_t available, std::size_t floor, const Bounds& b)
	{
		std::vector<std::size_t> chunks;
		while (available > floor)
		{
			const std::size_t take = std::min<std::size_t>(available, std::max<uint32_t>(b.max_samples, 1));
			chunks.push_back(take);
			available -= take;
		}
		return chunks;
	}
}
