#include "audio-vol.hpp"
#include "gpu-context.hpp"

#include <cstring>

namespace processor
{
	static std::vector<infra::Processor::Pin_attribute> io_pins()
	{
		return {
			{"output", "Output", typeid(Audio_stream), false, [] { return std::make_shared<Audio_stream>(); }},
			{"input", "Input", typeid(Audio_stream), true, [] { return std::make_shared<Audio_stream>(); }}
		};
	}

	infra::Processor::Info Audio_vol::get_processor_info()
	{
		return {"audio_volume_adjust", "Adjust Volume", false, [] { return std::unique_ptr<infra::Processor>(new Audio_vol); },
				"Audio Volume Adjuster (MI355X)"};
	}

	std::vector<infra::Processor::Pin_attribute> Audio_vol::get_pin_attributes() const { return io_pins(); }

	// same control flow as audio-vol.cpp:102-250; only the inner loop (:188-244) is a GPU call
	void Audio_vol::process_payload(
		const std::map<std::string, std::shared_ptr<infra::Processor::Product>>& input,
		const std::map<std::string, std::set<std::shared_ptr<infra::Processor::Product>>>& output,
		const std::atomic<bool>& stop_token,
		std::any&
	)
	{
		gpu::Node node;  // this node's context (own stream; device: gpu::pick_device): first local, destroyed last
		const auto input_item_optional = infra::get_input_item<Audio_stream>(input, "input");
		const auto output_item = infra::get_output_item<Audio_stream>(output, "output");
		if (!input_item_optional.has_value())
			throw Runtime_error(
				"Volume adjust processor has no input",
				"Volume adjust processor requires an audio stream input to function properly.",
				"Input item 'input' not found"
			);
		auto& input_item = input_item_optional.value().get();

		auto push_frame = [&stop_token, &output_item](const std::shared_ptr<Audio_frame>& frame)
		{
			for (auto& channel : output_item)
			{
				if (stop_token) return;
				while (channel->try_push(frame) != channel_op_status::success)
				{
					if (stop_token) return;
					nae_fiber::this_fiber::yield();
				}
			}
		};

		nae_ctx* ctx = gpu::context();
		gpu::Device_buffer d_src, d_dst;
		gpu::Pinned_buffer h_src, h_dst;
		batch_stats = {};

		// Batching: one launch per frame is launch-bound (a 1152-sample frame is 9 KB), so every
		// frame that is ALREADY waiting in the input stream (at most its capacity, 16) joins the batch as long as it
		// has the first frame's format and channel count.  The gain is elementwise, so the batch is one flat array of
		// samples: one upload sequence, one launch, one wait, then the frames are cut back out with their own
		// nb_samples / pts.  Nothing waits for more input: latency is that of the per-frame loop.
		constexpr size_t max_batch = 16;
		std::shared_ptr<const Audio_frame> held;  // popped, but of another format: opens the next batch
		struct Slot { std::shared_ptr<const Audio_frame> src; std::shared_ptr<Audio_frame> dst; size_t offset[2]; size_t plane_bytes; int planes; };

		while (!stop_token)
		{
			std::vector<Slot> batch;
			std::shared_ptr<const Audio_frame> first = std::move(held);
			held.reset();
			if (!first)
			{
				const auto pop_result = input_item.try_pop();
				if (!pop_result.has_value())
				{
					if (pop_result.error() == channel_op_status::empty)
					{
						if (input_item.eof()) break;
						nae_fiber::this_fiber::yield();
						continue;
					}
					else if (pop_result.error() == channel_op_status::closed)
						THROW_LOGIC_ERROR("Unexpected channel closed in Audio_vol::process_payload");
				}
				first = pop_result.value();
			}
			const int format = first->data()->format;
			const int ch = first->data()->ch_layout.nb_channels;
			if (ch != 1 && ch != 2)
				throw Runtime_error("Invalid channel count", "Only mono and stereo audio are supported.", infra::fmt("Got %d channels", ch));
			const int bps = bytes_per_sample(format);
			if (bps == 0 || format == AV_SAMPLE_FMT_DBL)
				throw Runtime_error("Audio format is not support", "Audio volume processor requires an audio format properly.", "Include FLT, S16, S32");
			const bool planar = sample_fmt_is_planar(format);

			size_t total = 0;  // bytes staged so far; every plane starts on a 256-byte boundary
			auto add = [&](std::shared_ptr<const Audio_frame> frame)
			{
				const Frame_data& src_frame = *frame->data();
				Slot slot;
				slot.dst = std::make_shared<Audio_frame>();
				Frame_data* out_frame = slot.dst->data();
				out_frame->sample_rate = src_frame.sample_rate;
				out_frame->format = src_frame.format;
				out_frame->nb_samples = src_frame.nb_samples;
				out_frame->ch_layout = src_frame.ch_layout;
				out_frame->pts = src_frame.pts;
				out_frame->time_base = src_frame.time_base;
				frame_get_buffer(out_frame, 32);
				slot.planes = planar ? ch : 1;
				slot.plane_bytes = (size_t)src_frame.nb_samples * bps * (planar ? 1 : ch);
				for (int p = 0; p < slot.planes; p++)
				{
					slot.offset[p] = total;
					total += (slot.plane_bytes + 255) / 256 * 256;
				}
				slot.src = std::move(frame);
				batch.push_back(std::move(slot));
			};
			add(std::move(first));
			while (batch.size() < max_batch)
			{
				const auto more = input_item.try_pop();
				if (!more.has_value()) break;  // nothing else is waiting (empty / closed are handled by the next round)
				const Frame_data* f = more.value()->data();
				if (f->format != format || f->ch_layout.nb_channels != ch)
				{
					held = more.value();
					break;
				}
				add(more.value());
			}

			auto* s = static_cast<uint8_t*>(d_src.reserve(total));
			auto* d = static_cast<uint8_t*>(d_dst.reserve(total));
			// the batch goes up and comes back as ONE asynchronous copy each, through page-locked staging (gpu::Pinned_buffer)
			auto* hs = static_cast<uint8_t*>(h_src.reserve(total));
			auto* hd = static_cast<uint8_t*>(h_dst.reserve(total));
			for (const Slot& slot : batch)
				for (int p = 0; p < slot.planes; p++) std::memcpy(hs + slot.offset[p], slot.src->data()->data[p], slot.plane_bytes);
			gpu::check(nae_memcpy_h2d(ctx, s, hs, total), "h2d");
			// the flat view of the batch as ONE packed mono plane of the sample type (the pad bytes between planes are
			// scaled too and never read back); `volume` is read once per batch, as the reference reads it once per frame
			const void* sp[1] = {s};
			void* dp[1] = {d};
			const int packed = planar ? (format == AV_SAMPLE_FMT_FLTP ? AV_SAMPLE_FMT_FLT : format == AV_SAMPLE_FMT_S16P ? AV_SAMPLE_FMT_S16 : AV_SAMPLE_FMT_S32) : format;
			gpu::check(nae_gain_frame(ctx, packed, sp, dp, total / bps, 1, volume), "nae_gain_frame");
			gpu::check(nae_memcpy_d2h(ctx, hd, d, total), "d2h");
			gpu::wait(stop_token);
			for (const Slot& slot : batch)
				for (int p = 0; p < slot.planes; p++) std::memcpy(slot.dst->data()->data[p], hd + slot.offset[p], slot.plane_bytes);
			batch_stats.rounds += batch.size();
			batch_stats.waits++;
			for (const Slot& slot : batch) push_frame(slot.dst);
		}
		for (auto& channel : output_item) channel->set_eof();  // audio-vol.cpp:249
	}
}
