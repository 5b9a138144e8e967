#include "audio-mix.hpp"
#include "gpu-context.hpp"

#include <algorithm>
#include <cmath>
#include <deque>
#include <list>

namespace processor
{
	namespace
	{
		constexpr int std_sample_rate = 48000;  // config::processor::audio_amix::std_sample_rate (config.hpp:63)

		std::vector<infra::Processor::Pin_attribute> lr_pins()
		{
			return {
				{"output", "Output", typeid(Audio_stream), false, [] { return std::make_shared<Audio_stream>(); }},
				{"input_l", "Left", typeid(Audio_stream), true, [] { return std::make_shared<Audio_stream>(); }},
				{"input_r", "Right", typeid(Audio_stream), true, [] { return std::make_shared<Audio_stream>(); }}
			};
		}

		// the SwrContext of the reference (audio-amix.cpp:212-240): any supported input -> 48 kHz stereo FLTP, as
		// nae_swr (identity inputs are a bit copy; other rates / mono / integer formats are converted on the GPU,
		// unpinned versus FFmpeg — include/nae_gpu.h "N2 input conversion")
		struct Gpu_swr
		{
			nae_swr* h = nullptr;

			Gpu_swr() = default;
			Gpu_swr(const Gpu_swr&) = delete;
			Gpu_swr& operator=(const Gpu_swr&) = delete;
			Gpu_swr(Gpu_swr&& o) noexcept : h(o.h) { o.h = nullptr; }
			~Gpu_swr() { if (h) nae_swr_destroy(h); }

			void open(const Frame_data& f)
			{
				if (h) return;
				const int rc = nae_swr_create(gpu::context(), f.format, f.sample_rate, f.ch_layout.nb_channels, std_sample_rate, &h);
				if (rc != NAE_OK)
					throw infra::Processor::Runtime_error(
						"Failed to initialize software resampler",
						"Cannot start the audio resampling process. Internal error may have occurred.",
						infra::fmt("nae_swr_create: %s (rate %d, channels %d, format %d)", nae_last_error(gpu::context()),
								   f.sample_rate, f.ch_layout.nb_channels, f.format)
					);
			}
			// swr_convert(ctx, out, S, in, n): out_* are zero-filled by the caller; returns the frames delivered
			int convert(const Frame_data* f, float* out_l, float* out_r, int S)
			{
				if (!h) return 0;  // an input that ended before its first frame contributes silence
				size_t got = 0;
				const int rc = f ? nae_swr_convert_host(h, reinterpret_cast<const void* const*>(f->data), f->nb_samples, out_l, out_r, S, &got)
								 : nae_swr_convert_host(h, nullptr, 0, out_l, out_r, S, &got);
				if (rc != NAE_OK)
					throw infra::Processor::Runtime_error(
						"Software resampler failed", "Cannot convert audio sample rate or format. Internal error may have occurred.",
						infra::fmt("nae_swr_convert_host returned error %d: %s", rc, nae_last_error(gpu::context()))
					);
				return (int)got;
			}
		};

		void push_to_all(const std::set<std::shared_ptr<Audio_stream>>& outs, const std::shared_ptr<Audio_frame>& frame,
						 const std::atomic<bool>& stop_token)
		{
			for (auto& channel : outs)
			{
				if (stop_token) return;
				while (channel->try_push(frame) != channel_op_status::success)
				{
					if (stop_token) return;
					nae_fiber::this_fiber::yield();
				}
			}
		}

		std::shared_ptr<Audio_frame> new_fltp_frame(int S, double time_seconds)
		{
			auto frame = std::make_shared<Audio_frame>();
			Frame_data* f = frame->data();
			f->nb_samples = S;
			f->ch_layout.nb_channels = 2;
			f->sample_rate = std_sample_rate;
			f->format = AV_SAMPLE_FMT_FLTP;
			f->pts = (int64_t)(time_seconds * 1000000);
			f->time_base = {1, 1000000};
			frame_get_buffer(f, 32);
			return frame;
		}
	}

	// ------------------------------------------------------------------------------------------ Audio_amix
	infra::Processor::Info Audio_amix::get_processor_info()
	{
		return {"audio_amix", "Audio Amix", false, [] { return std::unique_ptr<infra::Processor>(new Audio_amix); },
				"Multi-Channel Audio Mixer (MI355X)"};
	}

	std::vector<infra::Processor::Pin_attribute> Audio_amix::get_pin_attributes() const
	{
		std::vector<infra::Processor::Pin_attribute> pins;
		pins.push_back({"output", "Output", typeid(Audio_stream), false, [] { return std::make_shared<Audio_stream>(); }});
		for (int i = 0; i < input_num; i++)
			pins.push_back({infra::fmt("input_%d", i + 1), infra::fmt("Input %d", i + 1), typeid(Audio_stream), true,
							[] { return std::make_shared<Audio_stream>(); }});
		return pins;
	}

	Json::Value Audio_amix::serialize() const
	{
		Json::Value value;
		value["input_num"] = input_num;
		for (int i = 0; i < input_num; i++)
		{
			value[infra::fmt("volumes%d", i)] = volumes[i];
			value[infra::fmt("locks%d", i)] = (bool)locks[i];
		}
		return value;
	}

	void Audio_amix::deserialize(const Json::Value& value)
	{
		if (!value.isMember("input_num"))
			throw Runtime_error(
				"Failed to deserialize JSON file",
				"Audio_bimix failed to serialize the JSON input because of missing or invalid fields.",
				"Wrong field: input_num"
			);
		input_num = value["input_num"].asInt();
		locks.clear();
		volumes.clear();
		for (int i = 0; i < input_num; i++)
		{
			volumes.push_back(value[infra::fmt("volumes%d", i)].asFloat());
			locks.push_back(value[infra::fmt("locks%d", i)].asBool());
		}
	}

	// control flow of audio-amix.cpp:86-324; the mix loop :293-307 is nae_amix_f32
	void Audio_amix::process_payload(
		const std::map<std::string, std::shared_ptr<infra::Processor::Product>>& input,
		const std::map<std::string, std::set<std::shared_ptr<infra::Processor::Product>>>& output,
		const std::atomic<bool>& stop_token,
		std::any&
	)
	{
		const int input_num = this->input_num;
		if (input_num < 1 || input_num > 16) throw Runtime_error("Invalid input count", "The mixer supports 1 to 16 inputs.", infra::fmt("input_num = %d", input_num));
		if ((int)volumes.size() != input_num)  // the reference indexes an empty vector here (audio-amix.cpp:302) unless draw_content/deserialize ran
			throw Runtime_error("Audio Mixer has no volumes", "Deserialize the node (input_num, volumes{i}, locks{i}) before running it.", "volumes.size() != input_num");
		int count = 0;
		double time_seconds = 0;
		std::vector<bool> eofs(input_num, false);
		std::vector<std::vector<std::shared_ptr<const Audio_frame>>> buffers(input_num);
		std::vector<std::reference_wrapper<Audio_stream>> input_items;
		for (int i = 0; i < input_num; i++)
		{
			const auto try_item = infra::get_input_item<Audio_stream>(input, infra::fmt("input_%d", i + 1));
			if (!try_item.has_value())
				throw Runtime_error(
					"Audio Mixer processor has no input",
					"Audio Mixer processor requires an audio stream input to function properly.",
					infra::fmt("Input item 'input_%d' not found", i + 1)
				);
			input_items.emplace_back(try_item.value());
		}
		const auto output_item = infra::get_output_item<Audio_stream>(output, "output");
		std::vector<Gpu_swr> resamplers(input_num);

		nae_ctx* ctx = gpu::context();
		gpu::Device_buffer d_in, d_out;
		std::vector<float> h_in;

		while (!stop_token)
		{
			for (int i = 0; i < input_num; i++)
			{
				auto pop_result = input_items[i].get().try_pop();
				if (!pop_result.has_value())
				{
					if (pop_result.error() == channel_op_status::empty)
					{
						if (input_items[i].get().eof()) eofs[i] = true;
					}
					else if (pop_result.error() == channel_op_status::closed)
						THROW_LOGIC_ERROR("Unexpected channel closed in Audio_amix::process_payload");
				}
				else
					buffers[i].push_back(pop_result.value());
			}
			bool check = false;
			count = 0;
			for (int i = 0; i < input_num; i++)
				if (buffers[i].empty() && !eofs[i])
				{
					nae_fiber::this_fiber::yield();
					check = true;
					break;
				}
			if (check) continue;

			std::vector<const Frame_data*> frames(input_num, nullptr);
			for (int i = 0; i < input_num; i++) frames[i] = buffers[i].empty() ? nullptr : buffers[i].front()->data();
			int S = std::numeric_limits<int>::max();
			for (auto* f : frames)
				if (f) S = std::min(S, f->nb_samples);
			if (S == std::numeric_limits<int>::max()) S = 1152;  // :195
			time_seconds += S / double(std_sample_rate);          // :199 (pts = END time of the frame)
			auto new_frame = new_fltp_frame(S, time_seconds);
			for (int i = 0; i < input_num; i++)
				if (frames[i]) resamplers[i].open(*frames[i]);  // :206-243 (first frame of each input)
			// per-input swr_convert into zero-filled planes [i][2][S]
			const size_t plane = ((size_t)S + 3) / 4 * 4;
			h_in.assign((size_t)input_num * 2 * plane, 0.0f);
			for (int i = 0; i < input_num; i++)
			{
				const int got = resamplers[i].convert(frames[i], &h_in[(2 * i) * plane], &h_in[(2 * i + 1) * plane], S);
				if (!frames[i] && got < S) count++;  // :290
			}
			float* di = static_cast<float*>(d_in.reserve(h_in.size() * sizeof(float)));
			float* dout = static_cast<float*>(d_out.reserve(2 * plane * sizeof(float)));
			gpu::check(nae_memcpy_h2d(ctx, di, h_in.data(), h_in.size() * sizeof(float)), "h2d");
			const float *inL[16], *inR[16];
			for (int i = 0; i < input_num; i++) { inL[i] = di + (2 * i) * plane; inR[i] = di + (2 * i + 1) * plane; }
			gpu::check(nae_amix_f32(ctx, inL, inR, volumes.data(), input_num, dout, dout + plane, S), "nae_amix_f32");
			gpu::check(nae_memcpy_d2h(ctx, new_frame->data()->data[0], dout, S * sizeof(float)), "d2h");
			gpu::check(nae_memcpy_d2h(ctx, new_frame->data()->data[1], dout + plane, S * sizeof(float)), "d2h");
			gpu::wait(stop_token);

			for (auto& b : buffers)
				if (!b.empty()) b.erase(b.begin());
			push_to_all(output_item, new_frame, stop_token);
			if (count == input_num) break;  // :320
		}
		for (auto& out : output_item) out->set_eof();
	}

	// ------------------------------------------------------------------------------------------ Audio_bimix (v1)
	infra::Processor::Info Audio_bimix::get_processor_info()
	{
		return {"audio_bimix", "Audio Bimix", false, [] { return std::unique_ptr<infra::Processor>(new Audio_bimix); },
				"Stereo Channel Mixer (MI355X)"};
	}
	std::vector<infra::Processor::Pin_attribute> Audio_bimix::get_pin_attributes() const { return lr_pins(); }

	Json::Value Audio_bimix::serialize() const
	{
		Json::Value value;
		value["bias"] = bias;
		return value;
	}

	void Audio_bimix::deserialize(const Json::Value& value)
	{
		if (!value.isMember("bias") || !value["bias"].isDouble())
			throw Runtime_error(
				"Failed to deserialize JSON file",
				"Audio_bimix failed to serialize the JSON input because of missing or invalid fields.",
				"Wrong field: bias"
			);
		bias = (float)value["bias"].asDouble();
		bias = std::clamp<float>(bias, -1, 1);
	}

	// control flow of audio-bimix.cpp:83-331; the loop :310-317 is nae_bimix_f32.
	// Two reference quirks are NOT reproduced: time_seconds starts uninitialised (:103) — 0 here; the right-side
	// flush count lands in convert_count_l (:294) — here each side keeps its own count.
	void Audio_bimix::process_payload(
		const std::map<std::string, std::shared_ptr<infra::Processor::Product>>& input,
		const std::map<std::string, std::set<std::shared_ptr<infra::Processor::Product>>>& output,
		const std::atomic<bool>& stop_token,
		std::any&
	)
	{
		std::vector<std::shared_ptr<const Audio_frame>> buf_l, buf_r;
		bool left_eof = false, right_eof = false;
		double time_seconds = 0;
		const auto input_item_optional_l = infra::get_input_item<Audio_stream>(input, "input_l");
		const auto input_item_optional_r = infra::get_input_item<Audio_stream>(input, "input_r");
		const auto output_item = infra::get_output_item<Audio_stream>(output, "output");
		if (!input_item_optional_l.has_value() || !input_item_optional_r.has_value())
			throw Runtime_error(
				"Audio Channel mix processor has no input",
				"Audio channel mix processor requires an audio stream input to function properly.",
				"Input item 'input' not found"
			);
		auto& input_item_l = input_item_optional_l.value().get();
		auto& input_item_r = input_item_optional_r.value().get();
		Gpu_swr resampler_l, resampler_r;
		nae_ctx* ctx = gpu::context();
		gpu::Device_buffer d_in, d_out;
		std::vector<float> h_in;

		while (!stop_token)
		{
			const auto pop_result_l = input_item_l.try_pop();
			if (!pop_result_l.has_value()) { if (pop_result_l.error() == channel_op_status::empty && input_item_l.eof()) left_eof = true; }
			else buf_l.push_back(pop_result_l.value());
			const auto pop_result_r = input_item_r.try_pop();
			if (!pop_result_r.has_value()) { if (pop_result_r.error() == channel_op_status::empty && input_item_r.eof()) right_eof = true; }
			else buf_r.push_back(pop_result_r.value());
			if ((buf_r.empty() && !right_eof) || (buf_l.empty() && !left_eof))
			{
				nae_fiber::this_fiber::yield();
				continue;
			}
			const Frame_data* frame_l = buf_l.empty() ? nullptr : buf_l.front()->data();
			const Frame_data* frame_r = buf_r.empty() ? nullptr : buf_r.front()->data();
			int S;  // :176-183 (the reference's if / if / else-if / else chain ends in 1152 unless exactly one side is present)
			if (frame_r && frame_l) S = std::min(frame_r->nb_samples, frame_l->nb_samples);
			if (!frame_r && frame_l) S = frame_l->nb_samples;
			else if (frame_r && !frame_l) S = frame_r->nb_samples;
			else S = 1152;
			time_seconds += S / double(48000);
			auto new_frame = new_fltp_frame(S, time_seconds);
			if (frame_l) resampler_l.open(*frame_l);
			if (frame_r) resampler_r.open(*frame_r);
			const size_t plane = ((size_t)S + 3) / 4 * 4;
			h_in.assign(4 * plane, 0.0f);
			const int convert_count_l = resampler_l.convert(frame_l, &h_in[0], &h_in[plane], S);
			const int convert_count_r = resampler_r.convert(frame_r, &h_in[2 * plane], &h_in[3 * plane], S);
			float* di = static_cast<float*>(d_in.reserve(h_in.size() * sizeof(float)));
			float* dout = static_cast<float*>(d_out.reserve(2 * plane * sizeof(float)));
			gpu::check(nae_memcpy_h2d(ctx, di, h_in.data(), h_in.size() * sizeof(float)), "h2d");
			gpu::check(nae_bimix_f32(ctx, di, di + plane, di + 2 * plane, di + 3 * plane, bias, dout, dout + plane, S), "nae_bimix_f32");
			gpu::check(nae_memcpy_d2h(ctx, new_frame->data()->data[0], dout, S * sizeof(float)), "d2h");
			gpu::check(nae_memcpy_d2h(ctx, new_frame->data()->data[1], dout + plane, S * sizeof(float)), "d2h");
			gpu::wait(stop_token);
			if (frame_l) buf_l.erase(buf_l.begin());
			if (frame_r) buf_r.erase(buf_r.begin());
			push_to_all(output_item, new_frame, stop_token);
			if (convert_count_r == 0 && convert_count_l == 0) break;  // :327
		}
		for (auto& out : output_item) out->set_eof();
	}

	// ------------------------------------------------------------------------------------------ Audio_bimix_v2
	infra::Processor::Info Audio_bimix_v2::get_processor_info()
	{
		return {"audio_bimix_v2", "Audio Bimix V2", false, [] { return std::unique_ptr<infra::Processor>(new Audio_bimix_v2); },
				"Advanced Stereo Channel Mixer V2 (MI355X)"};
	}
	std::vector<infra::Processor::Pin_attribute> Audio_bimix_v2::get_pin_attributes() const { return lr_pins(); }

	// control flow of audio-bimix.cpp:475-877.  GPU calls: downmix (:624-627,717-720) and interleave with zero
	// fill (:797-803, :833-850, :736-742, :759-765).  The pts-alignment bookkeeping (:777-872) is host logic, kept
	// statement for statement.
	void Audio_bimix_v2::process_payload(
		const std::map<std::string, std::shared_ptr<infra::Processor::Product>>& input,
		const std::map<std::string, std::set<std::shared_ptr<infra::Processor::Product>>>& output,
		const std::atomic<bool>& stop_token,
		std::any&
	)
	{
		auto input_item_optional_l = infra::get_input_item<Audio_stream>(input, "input_l");
		auto input_item_optional_r = infra::get_input_item<Audio_stream>(input, "input_r");
		if (!input_item_optional_l.has_value() || !input_item_optional_r.has_value())
			throw Runtime_error(
				"Audio Channel mix processor has no input",
				"Audio channel mix processor requires an audio stream input to function properly.",
				"Input item 'input' not found"
			);
		auto& input_stream_l = input_item_optional_l.value().get();
		auto& input_stream_r = input_item_optional_r.value().get();
		auto output_stream = infra::get_output_item<Audio_stream>(output, "output");
		constexpr auto target_sample_rate = 48000;  // config::processor::audio_bimix::std_sample_rate

		struct Frame
		{
			std::vector<float> samples;
			double time_seconds = 0.0;
			double elapsed_seconds() const { return double(samples.size()) / target_sample_rate; }
			double end_time() const { return time_seconds + elapsed_seconds(); }
			void drop_samples(size_t count)
			{
				samples.erase(samples.begin(), samples.begin() + count);
				time_seconds += double(count) / target_sample_rate;
			}
		};
		std::list<Frame> frames_l, frames_r;
		bool have_l = false, have_r = false;
		double time_l = 0, time_r = 0;
		bool eof_l = false, eof_r = false;
		nae_ctx* ctx = gpu::context();
		gpu::Device_buffer d_a, d_b, d_out;

		// one side's intake: "resample" (identity) + mono downmix on the GPU
		Gpu_swr resampler_l, resampler_r;
		auto intake = [&](Audio_stream& stream, bool& eof, bool& have, double& t, std::list<Frame>& frames, Gpu_swr& resampler)
		{
			if (eof) return;
			const auto pop_result = stream.try_pop();
			if (!pop_result.has_value())
			{
				if (pop_result.error() == channel_op_status::empty && stream.eof()) eof = true;
				return;
			}
			const auto& data = *pop_result.value()->data();
			if (data.ch_layout.nb_channels != 2 && data.ch_layout.nb_channels != 1)
				throw Runtime_error("Invalid audio channel layout", "Audio channel layout must be stereo or mono.",
									infra::fmt("Invalid channel layout: %d", data.ch_layout.nb_channels));
			if (!have)
			{
				have = true;
				resampler.open(data);                   // :563-588
				t = data.pts * av_q2d(data.time_base);  // :589
			}
			// :594-604: resample into buffers of 2*nb_samples; the count delivered advances the clock
			std::vector<float> l(2 * (size_t)data.nb_samples + 1, 0.0f), r(2 * (size_t)data.nb_samples + 1, 0.0f);
			const int n = resampler.convert(&data, l.data(), r.data(), 2 * data.nb_samples);
			t += double(n) / target_sample_rate;  // :618 — the frame is stamped with its END time
			if (n == 0) return;
			Frame new_frame;
			new_frame.time_seconds = t;
			new_frame.samples.resize(n);
			float* da = static_cast<float*>(d_a.reserve(n * sizeof(float)));
			float* db = static_cast<float*>(d_b.reserve(n * sizeof(float)));
			float* dm = static_cast<float*>(d_out.reserve(2 * n * sizeof(float) + 64));
			gpu::check(nae_memcpy_h2d(ctx, da, l.data(), n * sizeof(float)), "h2d");
			gpu::check(nae_memcpy_h2d(ctx, db, r.data(), n * sizeof(float)), "h2d");
			gpu::check(nae_bimix2_downmix_f32(ctx, da, db, dm, n), "nae_bimix2_downmix_f32");
			gpu::check(nae_memcpy_d2h(ctx, new_frame.samples.data(), dm, n * sizeof(float)), "d2h");
			gpu::wait(stop_token);
			frames.emplace_back(std::move(new_frame));
		};

		auto make_audio_frame_flt_interleaved = [&](const float* earlier, const float* later, size_t unaligned,
													 size_t aligned, int earlier_offset, double time_seconds)
		{
			const size_t n = unaligned + aligned;
			auto frame = std::make_shared<Audio_frame>();  // :451-473
			Frame_data* data = frame->data();
			data->nb_samples = (int)n;
			data->ch_layout.nb_channels = 2;
			data->sample_rate = target_sample_rate;
			data->format = AV_SAMPLE_FMT_FLT;
			data->pts = (int64_t)(time_seconds * 1000000);
			data->time_base = {1, 1000000};
			frame_get_buffer(data, 32);
			if (n == 0) return frame;
			float* da = static_cast<float*>(d_a.reserve(n * sizeof(float)));
			float* db = static_cast<float*>(d_b.reserve((aligned ? aligned : 1) * sizeof(float)));
			float* dd = static_cast<float*>(d_out.reserve(2 * n * sizeof(float) + 64));
			gpu::check(nae_memcpy_h2d(ctx, da, earlier, n * sizeof(float)), "h2d");
			if (aligned) gpu::check(nae_memcpy_h2d(ctx, db, later, aligned * sizeof(float)), "h2d");
			gpu::check(nae_bimix2_interleave_f32(ctx, dd, da, aligned ? db : nullptr, unaligned, aligned, earlier_offset), "nae_bimix2_interleave_f32");
			gpu::check(nae_memcpy_d2h(ctx, data->data[0], dd, 2 * n * sizeof(float)), "d2h");
			gpu::wait(stop_token);
			return frame;
		};

		while (!stop_token)
		{
			nae_fiber::this_fiber::yield();
			intake(input_stream_l, eof_l, have_l, time_l, frames_l, resampler_l);
			intake(input_stream_r, eof_r, have_r, time_r, frames_r, resampler_r);

			if (frames_l.empty() && frames_r.empty() && eof_l && eof_r) break;
			if (frames_r.empty() && eof_r)  // right ended: :732-752
			{
				if (frames_l.empty()) continue;
				auto& f = frames_l.front();
				push_to_all(output_stream, make_audio_frame_flt_interleaved(f.samples.data(), nullptr, f.samples.size(), 0, 0, f.time_seconds), stop_token);
				frames_l.pop_front();
				continue;
			}
			if (frames_l.empty() && eof_l)  // left ended: :755-775
			{
				if (frames_r.empty()) continue;
				auto& f = frames_r.front();
				push_to_all(output_stream, make_audio_frame_flt_interleaved(f.samples.data(), nullptr, f.samples.size(), 0, 1, f.time_seconds), stop_token);
				frames_r.pop_front();
				continue;
			}
			while (!frames_l.empty() && !frames_r.empty() && !stop_token)  // :777-872
			{
				const bool left_eariler = frames_l.front().time_seconds < frames_r.front().time_seconds;
				const int eariler_offset = left_eariler ? 0 : 1;
				auto& eariler_stream = left_eariler ? frames_l : frames_r;
				auto& later_stream = left_eariler ? frames_r : frames_l;
				const double eariler_begin_time = eariler_stream.front().time_seconds;
				const double later_begin_time = later_stream.front().time_seconds;
				const double eariler_end_time = eariler_stream.front().end_time();
				const double later_end_time = later_stream.front().end_time();
				if (eariler_end_time <= later_begin_time)
				{
					auto& f = eariler_stream.front();
					push_to_all(output_stream, make_audio_frame_flt_interleaved(f.samples.data(), nullptr, f.samples.size(), 0, eariler_offset, eariler_begin_time), stop_token);
					eariler_stream.pop_front();
					continue;
				}
				const double frame_end_time = std::min(eariler_end_time, later_end_time);
				const auto unaligned_samples = static_cast<size_t>(std::round((later_begin_time - eariler_begin_time) * target_sample_rate));
				auto aligned_samples = static_cast<size_t>(std::round((frame_end_time - later_begin_time) * target_sample_rate));
				aligned_samples = std::min(aligned_samples, eariler_stream.front().samples.size() - unaligned_samples);
				aligned_samples = std::min(aligned_samples, later_stream.front().samples.size());
				auto frame = make_audio_frame_flt_interleaved(
					eariler_stream.front().samples.data(), later_stream.front().samples.data(), unaligned_samples,
					aligned_samples, eariler_offset, eariler_begin_time
				);
				if (eariler_end_time <= later_end_time)
				{
					eariler_stream.pop_front();
					later_stream.front().drop_samples(aligned_samples);
				}
				else
				{
					later_stream.pop_front();
					eariler_stream.front().drop_samples(unaligned_samples + aligned_samples);
				}
				if (!eariler_stream.empty() && eariler_stream.front().samples.empty()) eariler_stream.pop_front();
				if (!later_stream.empty() && later_stream.front().samples.empty()) later_stream.pop_front();
				push_to_all(output_stream, frame, stop_token);
			}
		}
		for (auto& stream : output_stream) stream->set_eof();
	}
}
