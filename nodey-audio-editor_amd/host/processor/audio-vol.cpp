#include "audio-vol.hpp"
#include "gpu-context.hpp"

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

		while (!stop_token)
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
			const auto& src_frame = *pop_result.value()->data();
			std::shared_ptr<Audio_frame> dst_frame = std::make_shared<Audio_frame>();
			Frame_data* out_frame = dst_frame->data();
			out_frame->sample_rate = src_frame.sample_rate;
			out_frame->format = src_frame.format;
			out_frame->nb_samples = src_frame.nb_samples;
			out_frame->ch_layout = src_frame.ch_layout;
			out_frame->pts = src_frame.pts;
			out_frame->time_base = src_frame.time_base;

			const int ch = src_frame.ch_layout.nb_channels;
			if (ch != 1 && ch != 2)
				throw Runtime_error("Invalid channel count", "Only mono and stereo audio are supported.", infra::fmt("Got %d channels", ch));
			const int bps = bytes_per_sample(src_frame.format);
			if (bps == 0 || src_frame.format == AV_SAMPLE_FMT_DBL)
				throw Runtime_error("Audio format is not support", "Audio volume processor requires an audio format properly.", "Include FLT, S16, S32");
			frame_get_buffer(out_frame, 32);

			// stage: planes packed back to back at 256-byte multiples so the 16-byte kernel path is taken
			const bool planar = sample_fmt_is_planar(src_frame.format);
			const int planes = planar ? ch : 1;
			const size_t plane_bytes = (size_t)src_frame.nb_samples * bps * (planar ? 1 : ch);
			const size_t stride = (plane_bytes + 255) / 256 * 256;
			auto* s = static_cast<uint8_t*>(d_src.reserve(stride * planes));
			auto* d = static_cast<uint8_t*>(d_dst.reserve(stride * planes));
			const void* sp[2] = {s, s + stride};
			void* dp[2] = {d, d + stride};
			for (int p = 0; p < planes; p++) gpu::check(nae_memcpy_h2d(ctx, s + p * stride, src_frame.data[p], plane_bytes), "h2d");
			const int rc = nae_gain_frame(ctx, src_frame.format, sp, dp, src_frame.nb_samples, ch, volume);  // volume read live per frame
			gpu::check(rc, "nae_gain_frame");
			for (int p = 0; p < planes; p++) gpu::check(nae_memcpy_d2h(ctx, out_frame->data[p], d + p * stride, plane_bytes), "d2h");
			gpu::wait(stop_token);
			push_frame(dst_frame);
		}
		for (auto& channel : output_item) channel->set_eof();  // audio-vol.cpp:249
	}
}
