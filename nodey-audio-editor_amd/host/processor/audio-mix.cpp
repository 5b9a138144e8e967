#include "audio-mix.hpp"
#include "bimix-align.hpp"
#include <deque>
#include "gpu-context.hpp"

#include <cstring>

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
			// the same, queued: DEVICE planes (zero-filled by the caller), nothing waited for; the frame stays alive until the
			// caller has waited for the stream
			int convert_queued(const Frame_data* f, float* dev_l, float* dev_r, int S)
			{
				if (!h) return 0;
				size_t got = 0;
				const int rc = f ? nae_swr_convert(h, reinterpret_cast<const void* const*>(f->data), f->nb_samples, dev_l, dev_r, S, &got)
								 : nae_swr_convert(h, nullptr, 0, dev_l, dev_r, S, &got);
				if (rc != NAE_OK)
					throw infra::Processor::Runtime_error(
						"Software resampler failed", "Cannot convert audio sample rate or format. Internal error may have occurred.",
						infra::fmt("nae_swr_convert returned error %d: %s", rc, nae_last_error(gpu::context()))
					);
				return (int)got;
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
		gpu::Node node;  // this node's context (own stream; device: gpu::pick_device): first local, destroyed last
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
		gpu::Pinned_buffer h_in, h_out;

		// Batching.  The reference takes one frame per input and round and mixes it; a 1152-sample round is
		// 9 KB per input, so on the GPU a round is all launch and wait.  Here every frame that is ALREADY waiting joins: the
		// rounds that can be formed from the buffered frames (at most 16) are planned first, then every round's input
		// conversions (nae_swr_convert: device planes, nothing waited for), its mix and its two downloads are queued, and the
		// stream is waited for once.  Round by round the arithmetic, the frame sizes, the time stamps and the end-of-stream
		// rule (:290, :320) are those of the one-round loop; nothing waits for input that has not arrived.
		constexpr size_t max_batch = 16;
		struct Round { int S; size_t plane, in_off, out_off; std::shared_ptr<Audio_frame> out; };

		bool finished = false;
		batch_stats = {};
		while (!stop_token && !finished)
		{
			for (int i = 0; i < input_num; i++)
				while (buffers[i].size() < max_batch)
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
						break;
					}
					buffers[i].push_back(pop_result.value());
				}
			bool check = false;
			for (int i = 0; i < input_num; i++)
				if (buffers[i].empty() && !eofs[i])
				{
					nae_fiber::this_fiber::yield();
					check = true;
					break;
				}
			if (check) continue;

			// plan: round b takes the b-th buffered frame of every input; an input that has ended contributes its converter's
			// remainder (frame == nullptr); a round with no frame at all (everything ended: the drain round, S = 1152) closes the batch
			std::vector<Round> rounds;
			size_t in_floats = 0, out_floats = 0;
			for (size_t b = 0; b < max_batch; b++)
			{
				bool any = false, ready = true;
				int S = std::numeric_limits<int>::max();
				for (int i = 0; i < input_num; i++)
				{
					if (buffers[i].size() > b) { any = true; S = std::min(S, buffers[i][b]->data()->nb_samples); }
					else if (!eofs[i]) ready = false;
				}
				if (!ready || (!any && b > 0)) break;
				if (!any) S = 1152;  // :195
				Round r;
				r.S = S;
				r.plane = ((size_t)S + 3) / 4 * 4;
				r.in_off = in_floats;
				r.out_off = out_floats;
				in_floats += (size_t)input_num * 2 * r.plane;
				out_floats += 2 * r.plane;
				rounds.push_back(std::move(r));
				if (!any) break;
			}

			// Fast path: the leading rounds in which EVERY input has a frame that needs no conversion — 48 kHz stereo float (packed or
			// planar, the format an input had in the batch's first round), all of one length S, no remainder waiting in a converter.
			// Their frames are copied into page-locked staging and go up as ONE asynchronous copy; ONE launch mixes all of these
			// rounds (rounds as "streams" of nae_amix_sig_f32: the arithmetic of nae_amix_f32, same bits).  Everything else — other
			// rates or formats, mono, rounds of unequal frame lengths, the drain rounds behind an ended input — takes the converter
			// path below, one round at a time, in a later turn if fast rounds lead the batch.
			size_t fast = 0;
			for (size_t b = 0; b < rounds.size(); b++)
			{
				bool ok = rounds[b].S == rounds[0].S;
				for (int i = 0; i < input_num && ok; i++)
				{
					if (buffers[i].size() <= b) { ok = false; break; }
					const Frame_data* f = buffers[i][b]->data();
					ok = f->sample_rate == std_sample_rate && f->ch_layout.nb_channels == 2 && f->nb_samples == rounds[0].S &&
						 (f->format == AV_SAMPLE_FMT_FLT || f->format == AV_SAMPLE_FMT_FLTP) && f->format == buffers[i][0]->data()->format &&
						 (resamplers[i].h == nullptr || nae_swr_buffered(resamplers[i].h) == 0);
				}
				if (!ok) break;
				fast++;
			}
			if (fast > 0) rounds.resize(fast);
			out_floats = 0;
			for (const Round& r : rounds) out_floats += 2 * r.plane;

			float* dout = static_cast<float*>(d_out.reserve(out_floats * sizeof(float)));
			float* hout = static_cast<float*>(h_out.reserve(out_floats * sizeof(float)));
			size_t done = 0;
			if (fast > 0)
			{
				const int S = rounds[0].S;
				const size_t plane = rounds[0].plane, slot = 2 * plane;      // one input frame: [S][2] packed or [2][plane] planar
				const size_t n_floats = fast * (size_t)input_num * slot;
				float* di = static_cast<float*>(d_in.reserve(n_floats * sizeof(float)));
				float* hin = static_cast<float*>(h_in.reserve(n_floats * sizeof(float)));
				std::vector<nae_sig> sigs(input_num);
				for (int i = 0; i < input_num; i++)
				{
					const bool planar = buffers[i][0]->data()->format == AV_SAMPLE_FMT_FLTP;
					sigs[i] = planar ? nae_sig{di + (size_t)i * slot, (size_t)input_num * slot, plane, 1} : nae_sig{di + (size_t)i * slot, (size_t)input_num * slot, 1, 2};
					for (size_t b = 0; b < fast; b++)
					{
						const Frame_data* f = buffers[i][b]->data();
						float* dst = hin + (b * (size_t)input_num + (size_t)i) * slot;
						if (planar)
						{
							std::memcpy(dst, f->data[0], (size_t)S * sizeof(float));
							std::memcpy(dst + plane, f->data[1], (size_t)S * sizeof(float));
						}
						else
							std::memcpy(dst, f->data[0], (size_t)S * 2 * sizeof(float));
					}
				}
				gpu::check(nae_memcpy_h2d(ctx, di, hin, n_floats * sizeof(float)), "h2d");
				const nae_sig out_sig{dout, slot, plane, 1};
				gpu::check(nae_amix_sig_f32(ctx, sigs.data(), volumes.data(), input_num, &out_sig, (size_t)S, fast), "nae_amix_sig_f32");
				for (Round& r : rounds)
				{
					time_seconds += r.S / double(std_sample_rate);          // :199 (pts = END time of the frame)
					r.out = new_fltp_frame(r.S, time_seconds);
					done++;
				}
			}
			else
			{
			float* di = static_cast<float*>(d_in.reserve(in_floats * sizeof(float)));
			if (in_floats) gpu::check(nae_memset(ctx, di, 0, in_floats * sizeof(float)), "nae_memset");  // zero-filled planes [round][i][2][S]
			for (Round& r : rounds)
			{
				const size_t b = done;
				std::vector<const Frame_data*> frames(input_num, nullptr);
				for (int i = 0; i < input_num; i++) frames[i] = buffers[i].size() > b ? buffers[i][b]->data() : nullptr;
				time_seconds += r.S / double(std_sample_rate);          // :199 (pts = END time of the frame)
				r.out = new_fltp_frame(r.S, time_seconds);
				for (int i = 0; i < input_num; i++)
					if (frames[i]) resamplers[i].open(*frames[i]);  // :206-243 (first frame of each input)
				count = 0;
				const float *inL[16], *inR[16];
				for (int i = 0; i < input_num; i++)
				{
					float* l = di + r.in_off + (size_t)(2 * i) * r.plane;
					float* rr = l + r.plane;
					const int got = resamplers[i].convert_queued(frames[i], l, rr, r.S);
					if (!frames[i] && got < r.S) count++;  // :290
					inL[i] = l;
					inR[i] = rr;
				}
				float* o = dout + r.out_off;
				gpu::check(nae_amix_f32(ctx, inL, inR, volumes.data(), input_num, o, o + r.plane, r.S), "nae_amix_f32");
				done++;
				if (count == input_num) { finished = true; break; }  // :320
			}
			}
			// all mixed rounds come down as ONE asynchronous copy into page-locked staging
			if (done) gpu::check(nae_memcpy_d2h(ctx, hout, dout, (rounds[done - 1].out_off + 2 * rounds[done - 1].plane) * sizeof(float)), "d2h");
			gpu::wait(stop_token);  // the buffered frames (the converter path's upload sources) are released only now
			for (size_t k = 0; k < done; k++)
			{
				std::memcpy(rounds[k].out->data()->data[0], hout + rounds[k].out_off, (size_t)rounds[k].S * sizeof(float));
				std::memcpy(rounds[k].out->data()->data[1], hout + rounds[k].out_off + rounds[k].plane, (size_t)rounds[k].S * sizeof(float));
			}
			batch_stats.rounds += done;
			batch_stats.waits++;

			for (auto& b : buffers) b.erase(b.begin(), b.begin() + std::min(done, b.size()));
			for (size_t k = 0; k < done; k++) push_to_all(output_item, rounds[k].out, stop_token);
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
		gpu::Node node;  // this node's context (own stream; device: gpu::pick_device): first local, destroyed last
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
		gpu::Pinned_buffer h_out;

		// batched like Audio_amix: the rounds that the waiting frames allow are planned, queued and waited for once
		constexpr size_t max_batch = 16;
		struct Round { int S; size_t plane, in_off, out_off; std::shared_ptr<Audio_frame> out; };
		auto intake = [&](Audio_stream& item, std::vector<std::shared_ptr<const Audio_frame>>& buf, bool& eof)
		{
			while (buf.size() < max_batch)
			{
				const auto pop_result = item.try_pop();
				if (!pop_result.has_value())
				{
					if (pop_result.error() == channel_op_status::empty && item.eof()) eof = true;
					break;
				}
				buf.push_back(pop_result.value());
			}
		};

		bool finished = false;
		batch_stats = {};
		while (!stop_token && !finished)
		{
			intake(input_item_l, buf_l, left_eof);
			intake(input_item_r, buf_r, right_eof);
			if ((buf_r.empty() && !right_eof) || (buf_l.empty() && !left_eof))
			{
				nae_fiber::this_fiber::yield();
				continue;
			}
			std::vector<Round> rounds;
			size_t in_floats = 0, out_floats = 0;
			for (size_t b = 0; b < max_batch; b++)
			{
				const Frame_data* frame_l = buf_l.size() > b ? buf_l[b]->data() : nullptr;
				const Frame_data* frame_r = buf_r.size() > b ? buf_r[b]->data() : nullptr;
				if ((!frame_l && !left_eof) || (!frame_r && !right_eof)) break;
				if (!frame_l && !frame_r && b > 0) break;
				int S;  // :176-183 (the reference's if / if / else-if / else chain ends in 1152 unless exactly one side is present)
				if (frame_r && frame_l) S = std::min(frame_r->nb_samples, frame_l->nb_samples);
				if (!frame_r && frame_l) S = frame_l->nb_samples;
				else if (frame_r && !frame_l) S = frame_r->nb_samples;
				else S = 1152;
				Round r;
				r.S = S;
				r.plane = ((size_t)S + 3) / 4 * 4;
				r.in_off = in_floats;
				r.out_off = out_floats;
				in_floats += 4 * r.plane;
				out_floats += 2 * r.plane;
				rounds.push_back(std::move(r));
				if (!frame_l && !frame_r) break;
			}
			float* di = static_cast<float*>(d_in.reserve(in_floats * sizeof(float)));
			float* dout = static_cast<float*>(d_out.reserve(out_floats * sizeof(float)));
			if (in_floats) gpu::check(nae_memset(ctx, di, 0, in_floats * sizeof(float)), "nae_memset");
			size_t done = 0;
			for (Round& r : rounds)
			{
				const size_t b = done;
				const Frame_data* frame_l = buf_l.size() > b ? buf_l[b]->data() : nullptr;
				const Frame_data* frame_r = buf_r.size() > b ? buf_r[b]->data() : nullptr;
				time_seconds += r.S / double(48000);
				r.out = new_fltp_frame(r.S, time_seconds);
				if (frame_l) resampler_l.open(*frame_l);
				if (frame_r) resampler_r.open(*frame_r);
				float* in = di + r.in_off;
				const int convert_count_l = resampler_l.convert_queued(frame_l, in, in + r.plane, r.S);
				const int convert_count_r = resampler_r.convert_queued(frame_r, in + 2 * r.plane, in + 3 * r.plane, r.S);
				float* o = dout + r.out_off;
				gpu::check(nae_bimix_f32(ctx, in, in + r.plane, in + 2 * r.plane, in + 3 * r.plane, bias, o, o + r.plane, r.S), "nae_bimix_f32");
				done++;
				if (convert_count_r == 0 && convert_count_l == 0) { finished = true; break; }  // :327
			}
			// all rounds come down as ONE asynchronous copy into page-locked staging (a copy into the pageable frames would block
			// this thread until the stream has reached it)
			float* hout = static_cast<float*>(h_out.reserve(out_floats * sizeof(float)));
			if (done) gpu::check(nae_memcpy_d2h(ctx, hout, dout, (rounds[done - 1].out_off + 2 * rounds[done - 1].plane) * sizeof(float)), "d2h");
			gpu::wait(stop_token);
			for (size_t k = 0; k < done; k++)
			{
				std::memcpy(rounds[k].out->data()->data[0], hout + rounds[k].out_off, (size_t)rounds[k].S * sizeof(float));
				std::memcpy(rounds[k].out->data()->data[1], hout + rounds[k].out_off + rounds[k].plane, (size_t)rounds[k].S * sizeof(float));
			}
			batch_stats.rounds += done;
			batch_stats.waits++;
			buf_l.erase(buf_l.begin(), buf_l.begin() + std::min(done, buf_l.size()));
			buf_r.erase(buf_r.begin(), buf_r.begin() + std::min(done, buf_r.size()));
			for (size_t k = 0; k < done; k++) push_to_all(output_item, rounds[k].out, stop_token);
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

	// Audio_bimix_v2 (audio-bimix.cpp:475-877): each input is converted to 48 kHz, mixed down to mono on the GPU
	// (:624-627, :717-720) and queued with its presentation time; the two queues are then lined up on those times
	// (bimix-align.hpp) and every step leaves as one interleaved stereo frame built on the GPU — the side that is not playing
	// is silent (:797-803, :833-850, :736-742, :759-765).
	void Audio_bimix_v2::process_payload(
		const std::map<std::string, std::shared_ptr<infra::Processor::Product>>& input,
		const std::map<std::string, std::set<std::shared_ptr<infra::Processor::Product>>>& output,
		const std::atomic<bool>& stop_token,
		std::any&
	)
	{
		gpu::Node node;  // this node's context (own stream; device: gpu::pick_device): first local, destroyed last
		auto in_l = infra::get_input_item<Audio_stream>(input, "input_l");
		auto in_r = infra::get_input_item<Audio_stream>(input, "input_r");
		if (!in_l.has_value() || !in_r.has_value())
			throw Runtime_error(
				"Audio Channel mix processor has no input",
				"Audio channel mix processor requires an audio stream input to function properly.",
				"Input item 'input' not found"
			);
		auto output_stream = infra::get_output_item<Audio_stream>(output, "output");
		constexpr int rate = 48000;  // config::processor::audio_bimix::std_sample_rate
		nae_ctx* ctx = gpu::context();
		gpu::Device_buffer d_a, d_b, d_out;

		// a mono block waiting to be placed: samples [at, mono.size()) are pending, the first of them plays at `begin`
		struct Pending
		{
			std::vector<float> mono;
			size_t at = 0;
			double begin = 0.0;
			bimix::Span span() const { return {begin, mono.size() - at}; }
			const float* data() const { return mono.data() + at; }
			void play(size_t n) { at += n; begin += double(n) / rate; }
		};
		struct Side
		{
			Audio_stream& stream;
			Gpu_swr resampler;
			std::deque<Pending> queue;
			bool opened = false, ended = false;
			double clock = 0.0;  // presentation time behind the last sample delivered by the resampler
			explicit Side(Audio_stream& s) : stream(s) {}
		};
		Side side[2] = {Side(in_l.value().get()), Side(in_r.value().get())};

		// Batching as in the other mixers: everything that is waiting is queued on the GPU and waited for once — on the intake side
		// every frame already in an input stream (conversion, mono mix-down and download per frame, one wait per side and round),
		// on the output side every frame the alignment loop can form from the queued blocks (upload, interleave, download per
		// frame, one wait per round).
		constexpr size_t max_batch = 16;
		batch_stats = {};

		// take the waiting frames off an input: convert, advance the side's clock by what the converter delivered (:594-618: a
		// block is stamped with the time BEHIND it), mix down to mono on the GPU and queue the blocks
		gpu::Device_buffer d_l, d_r, d_m;
		gpu::Pinned_buffer h_m, h_frames;
		auto intake = [&](Side& sd)
		{
			if (sd.ended) return;
			std::vector<std::shared_ptr<const Audio_frame>> frames;
			while (frames.size() < max_batch)
			{
				const auto popped = sd.stream.try_pop();
				if (!popped.has_value())
				{
					if (popped.error() == channel_op_status::empty && sd.stream.eof()) sd.ended = frames.empty();
					break;
				}
				frames.push_back(popped.value());
			}
			if (frames.empty()) return;
			size_t room_total = 0;
			for (const auto& f : frames)
			{
				const auto& data = *f->data();
				if (data.ch_layout.nb_channels != 2 && data.ch_layout.nb_channels != 1)
					throw Runtime_error("Invalid audio channel layout", "Audio channel layout must be stereo or mono.",
										infra::fmt("Invalid channel layout: %d", data.ch_layout.nb_channels));
				room_total += (2 * (size_t)data.nb_samples + 3) / 4 * 4;
			}
			float* dl = static_cast<float*>(d_l.reserve(room_total * sizeof(float)));
			float* dr = static_cast<float*>(d_r.reserve(room_total * sizeof(float)));
			float* dm = static_cast<float*>(d_m.reserve(room_total * sizeof(float)));
			float* hm = static_cast<float*>(h_m.reserve(room_total * sizeof(float)));  // page-locked: the downloads do not block this thread
			std::vector<Pending> staged;
			std::vector<size_t> staged_off;
			staged.reserve(frames.size());
			size_t off = 0;
			for (const auto& f : frames)
			{
				const auto& data = *f->data();
				if (!sd.opened)
				{
					sd.opened = true;
					sd.resampler.open(data);
					sd.clock = data.pts * av_q2d(data.time_base);
				}
				const size_t room = 2 * (size_t)data.nb_samples;
				const int n = sd.resampler.convert_queued(&data, dl + off, dr + off, (int)room);
				sd.clock += double(n) / rate;
				if (n > 0)
				{
					Pending block;
					block.begin = sd.clock;
					block.mono.resize(n);
					gpu::check(nae_bimix2_downmix_f32(ctx, dl + off, dr + off, dm + off, n), "nae_bimix2_downmix_f32");
					staged.emplace_back(std::move(block));
					staged_off.push_back(off);
					gpu::check(nae_memcpy_d2h(ctx, hm + off, dm + off, n * sizeof(float)), "d2h");
				}
				off += (room + 3) / 4 * 4;
			}
			gpu::wait(stop_token);  // the frames (upload sources) are released only now
			for (size_t k = 0; k < staged.size(); k++) std::memcpy(staged[k].mono.data(), hm + staged_off[k], staged[k].mono.size() * sizeof(float));
			batch_stats.rounds += frames.size();
			batch_stats.waits++;
			for (Pending& block : staged) sd.queue.emplace_back(std::move(block));
		};

		// output frames queued on the GPU since the last flush; the blocks their uploads read from stay alive in `retired`
		struct Job { std::shared_ptr<Audio_frame> frame; size_t host_off = 0, floats = 0; };
		std::vector<Job> jobs;
		std::vector<Pending> retired;
		size_t job_floats_a = 0, job_floats_b = 0, job_floats_out = 0, round_cap = 0;
		auto flush = [&]()
		{
			if (jobs.empty()) { retired.clear(); return; }
			gpu::wait(stop_token);
			batch_stats.rounds += jobs.size();
			batch_stats.waits++;
			for (Job& j : jobs)
			{
				if (j.floats) std::memcpy(j.frame->data()->data[0], static_cast<float*>(h_frames.reserve(0)) + j.host_off, j.floats * sizeof(float));
				push_to_all(output_stream, j.frame, stop_token);
			}
			jobs.clear();
			retired.clear();
			job_floats_a = job_floats_b = job_floats_out = 0;
		};
		// capacity for one round of output frames: every pending sample is played at most once per side
		auto reserve_round = [&]()
		{
			size_t pending = 64;
			for (const Side& sd : side)
				for (const Pending& q : sd.queue) pending += (q.span().count + 3) / 4 * 4 + 4;
			d_a.reserve(pending * sizeof(float));
			d_b.reserve(pending * sizeof(float));
			d_out.reserve(2 * pending * sizeof(float) + 64);
			h_frames.reserve(2 * pending * sizeof(float) + 64);
			round_cap = pending;
		};

		// one output frame: `solo` samples of side `first` alone, then `both` samples of the two sides (queued, not waited for)
		auto emit = [&](const float* first_side, const float* other_side, size_t solo, size_t both, int first, double begin)
		{
			const size_t n = solo + both;
			auto frame = std::make_shared<Audio_frame>();
			Frame_data* data = frame->data();
			data->nb_samples = (int)n;
			data->ch_layout.nb_channels = 2;
			data->sample_rate = rate;
			data->format = AV_SAMPLE_FMT_FLT;
			data->pts = (int64_t)(begin * 1000000);
			data->time_base = {1, 1000000};
			frame_get_buffer(data, 32);
			if (n)
			{
				float* da = d_a.as<float>() + job_floats_a;
				float* db = d_b.as<float>() + job_floats_b;
				float* dd = d_out.as<float>() + job_floats_out;
				gpu::check(nae_memcpy_h2d(ctx, da, first_side, n * sizeof(float)), "h2d");
				if (both) gpu::check(nae_memcpy_h2d(ctx, db, other_side, both * sizeof(float)), "h2d");
				gpu::check(nae_bimix2_interleave_f32(ctx, dd, da, both ? db : nullptr, solo, both, first), "nae_bimix2_interleave_f32");
				gpu::check(nae_memcpy_d2h(ctx, static_cast<float*>(h_frames.reserve(0)) + job_floats_out, dd, 2 * n * sizeof(float)), "d2h");
				jobs.push_back({frame, job_floats_out, 2 * n});
				job_floats_a += (n + 3) / 4 * 4;
				job_floats_b += (both + 3) / 4 * 4;
				job_floats_out += (2 * n + 3) / 4 * 4;
			}
			else
				jobs.push_back({frame, 0, 0});
		};
		auto emit_alone = [&](int which)
		{
			reserve_round();
			Pending& p = side[which].queue.front();
			emit(p.data(), nullptr, p.span().count, 0, which, p.begin);
			retired.emplace_back(std::move(p));
			side[which].queue.pop_front();
			flush();
		};

		while (!stop_token)
		{
			nae_fiber::this_fiber::yield();
			intake(side[0]);
			intake(side[1]);
			const bool idle[2] = {side[0].queue.empty(), side[1].queue.empty()};
			if (idle[0] && idle[1] && side[0].ended && side[1].ended) break;
			// one input has ended and is drained: the other plays on alone, block by block
			if (idle[1] && side[1].ended) { if (!idle[0]) emit_alone(0); continue; }
			if (idle[0] && side[0].ended) { if (!idle[1]) emit_alone(1); continue; }
			reserve_round();
			while (!side[0].queue.empty() && !side[1].queue.empty() && !stop_token)
			{
				Pending* p[2] = {&side[0].queue.front(), &side[1].queue.front()};
				// (a frame takes at most both blocks' samples: if the staging buffers could not hold it, what is queued goes out first)
				if (job_floats_a + p[0]->span().count + p[1]->span().count + 16 > round_cap) flush();
				const bimix::Step st = bimix::align_step(p[0]->span(), p[1]->span(), rate);
				const int a = st.first, b = 1 - st.first;
				const float* fa = p[a]->data();
				const float* fb = p[b]->data();
				const double begin = p[a]->begin;
				// the queues change before the frame is built: its sample pointers stay valid (a deque keeps the blocks it does not
				// remove in place, and the removed ones are moved to `retired`, which lives until the uploads have been waited for)
				for (int k = 0; k < 2; k++)
				{
					if (st.used_up[k]) { retired.emplace_back(std::move(*p[k])); side[k].queue.pop_front(); }
					else p[k]->play(st.played[k]);
				}
				for (int k = 0; k < 2; k++)
					if (!st.used_up[k] && p[k]->span().count == 0) { retired.emplace_back(std::move(*p[k])); side[k].queue.pop_front(); }
				emit(fa, fb, st.solo, st.both, a, begin);
			}
			flush();
		}
		for (auto& stream : output_stream) stream->set_eof();
	}
}
