#include "audio-stream.hpp"

namespace processor
{
	bool sample_fmt_is_planar(int fmt) { return fmt == AV_SAMPLE_FMT_S16P || fmt == AV_SAMPLE_FMT_S32P || fmt == AV_SAMPLE_FMT_FLTP; }

	int bytes_per_sample(int fmt)
	{
		switch (fmt)
		{
		case AV_SAMPLE_FMT_S16:
		case AV_SAMPLE_FMT_S16P: return 2;
		case AV_SAMPLE_FMT_S32:
		case AV_SAMPLE_FMT_S32P:
		case AV_SAMPLE_FMT_FLT:
		case AV_SAMPLE_FMT_FLTP: return 4;
		case AV_SAMPLE_FMT_DBL: return 8;
		default: return 0;
		}
	}

	int frame_get_buffer(Frame_data* f, int align)
	{
		const int bps = bytes_per_sample(f->format);
		const int ch = f->ch_layout.nb_channels;
		if (bps == 0 || ch < 1 || ch > 8 || f->nb_samples < 0) return -1;
		if (align <= 0) align = 32;
		const bool planar = sample_fmt_is_planar(f->format);
		const size_t planes = planar ? ch : 1;
		size_t plane_bytes = (size_t)f->nb_samples * bps * (planar ? 1 : ch);
		plane_bytes = (plane_bytes + align - 1) / align * align;
		f->storage.assign(planes * plane_bytes + align, 0);
		auto base = reinterpret_cast<uintptr_t>(f->storage.data());
		base = (base + align - 1) / align * align;
		for (size_t p = 0; p < 8; p++) f->data[p] = p < planes ? reinterpret_cast<uint8_t*>(base + p * plane_bytes) : nullptr;
		return 0;
	}

	// reference: src/processor/audio-stream.cpp:60-66
	channel_op_status Audio_stream::try_push(std::shared_ptr<const Audio_frame> frame)
	{
		if (channel.size() >= capacity) return channel_op_status::full;
		channel.push_back(std::move(frame));
		++buffered_frames;
		return channel_op_status::success;
	}

	// reference: src/processor/audio-stream.cpp:68-80
	Pop_result Audio_stream::try_pop()
	{
		if (channel.empty()) return channel_op_status::empty;
		auto frame = std::move(channel.front());
		channel.pop_front();
		--buffered_frames;
		return frame;
	}
}
