// processor/audio-stream.hpp — mirror of /root/reference/include/processor/audio-stream.hpp:22-83.
// Audio_frame wraps an AVFrame-shaped record (the fields the hot path reads: format, sample_rate, nb_samples,
// ch_layout.nb_channels, pts, time_base, data[] planes) without FFmpeg; Audio_stream is the bounded, never-blocking
// frame queue (capacity config::processor::audio_stream::buffer_size = 16, include/config.hpp:53).
#pragma once

#include "../infra/processor.hpp"

#include <atomic>
#include <cstdint>
#include <deque>
#include <memory>
#include <vector>

namespace config::processor::audio_stream { inline constexpr size_t buffer_size = 16; }

namespace processor
{
	// == AVSampleFormat numbering, so values pass through to the C ABI unchanged
	enum Sample_format : int
	{
		AV_SAMPLE_FMT_S16 = 1, AV_SAMPLE_FMT_S32 = 2, AV_SAMPLE_FMT_FLT = 3, AV_SAMPLE_FMT_DBL = 4,
		AV_SAMPLE_FMT_S16P = 6, AV_SAMPLE_FMT_S32P = 7, AV_SAMPLE_FMT_FLTP = 8
	};
	bool sample_fmt_is_planar(int fmt);
	int bytes_per_sample(int fmt);

	// AVFrame-shaped
	struct Frame_data
	{
		int format = -1;
		int sample_rate = 0;
		int nb_samples = 0;
		struct { int nb_channels = 0; } ch_layout;
		int64_t pts = 0;
		struct { int num = 0, den = 1; } time_base;
		uint8_t* data[8] = {};
		std::vector<uint8_t> storage;  // owns the planes (av_frame_get_buffer)
	};
	inline double av_q2d(decltype(Frame_data::time_base) q) { return q.num / (double)q.den; }
	// allocate planes for the frame's format / nb_samples / channels, each `align`-byte aligned
	// (the reference calls av_frame_get_buffer(frame, 32): audio-vol.cpp:173)
	int frame_get_buffer(Frame_data* frame, int align);

	class Audio_frame
	{
		std::unique_ptr<Frame_data> frame;

	  public:

		Audio_frame() : frame(std::make_unique<Frame_data>()) {}
		virtual ~Audio_frame() = default;
		Audio_frame(const Audio_frame&) = delete;
		Audio_frame& operator=(const Audio_frame&) = delete;

		Frame_data* data() { return frame.get(); }
		const Frame_data* data() const { return frame.get(); }
		Frame_data* operator->() { return frame.get(); }
		const Frame_data* operator->() const { return frame.get(); }
	};

	// boost::fibers::channel_op_status, the subset Audio_stream returns
	enum class channel_op_status { success, empty, full, closed };

	// std::expected<shared_ptr<const Audio_frame>, channel_op_status> as used at audio-vol.cpp:142-154
	class Pop_result
	{
		std::shared_ptr<const Audio_frame> frame;
		channel_op_status status;

	  public:

		Pop_result(std::shared_ptr<const Audio_frame> f) : frame(std::move(f)), status(channel_op_status::success) {}
		Pop_result(channel_op_status s) : status(s) {}
		bool has_value() const { return status == channel_op_status::success; }
		const std::shared_ptr<const Audio_frame>& value() const { return frame; }
		channel_op_status error() const { return status; }
	};

	class Audio_stream : public infra::Processor::Product
	{
		std::deque<std::shared_ptr<const Audio_frame>> channel;
		size_t capacity;
		std::atomic<size_t> buffered_frames = 0;
		std::atomic<bool> end_of_stream;

	  public:

		Audio_stream() : capacity(config::processor::audio_stream::buffer_size), end_of_stream(false) {}
		Audio_stream(const Audio_stream&) = delete;
		Audio_stream& operator=(const Audio_stream&) = delete;

		channel_op_status try_push(std::shared_ptr<const Audio_frame> frame);  // never blocks: success | full
		Pop_result try_pop();                                                  // never blocks: frame | empty
		bool eof() const { return end_of_stream.load(); }
		void set_eof() { end_of_stream.store(true); }
		size_t buffered_count() const { return buffered_frames.load(); }
	};

	// how a batching node's last run went: `rounds` units of the reference's one-per-iteration loop (frames mixed / frames put)
	// were served behind `waits` waits for the GPU; rounds / waits is the average batch
	struct Batch_stats
	{
		size_t rounds = 0, waits = 0;
	};
}
