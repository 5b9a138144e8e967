#include "audio-velocity.hpp"
#include "gpu-context.hpp"
#include "velocity-cadence.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace processor
{
	const char* algorithm_name(Stretch_algorithm a) { return a == Stretch_algorithm::Soundtouch ? "soundtouch" : "vocoder"; }

	namespace
	{
		Stretch_algorithm g_default_algorithm = Stretch_algorithm::Vocoder;
	}
	Stretch_algorithm default_stretch_algorithm() { return g_default_algorithm; }
	void set_default_stretch_algorithm(Stretch_algorithm a) { g_default_algorithm = a; }

	Stretch_algorithm algorithm_from_json(const Json::Value& value)
	{
		if (value.isMember("algorithm") && value["algorithm"].isString())
		{
			if (value["algorithm"].asString() == "soundtouch") return Stretch_algorithm::Soundtouch;
			if (value["algorithm"].asString() == "vocoder") return Stretch_algorithm::Vocoder;
		}
		return default_stretch_algorithm();   // no key: a project saved by the reference
	}

	namespace
	{
		std::vector<infra::Processor::Pin_attribute> io_pins()
		{
			return {
				{"output", "Output", typeid(Audio_stream), false, [] { return std::make_shared<Audio_stream>(); }},
				{"input", "Input", typeid(Audio_stream), true, [] { return std::make_shared<Audio_stream>(); }}
			};
		}

		// construct_audio_frame_float, audio-velocity.cpp:234-263 (time_us is a FLOAT there: 24-bit pts quirk kept)
		std::shared_ptr<Audio_frame> construct_audio_frame_float(const float* samples, size_t n_frames, int sample_rate,
																 int channel_count, float time_us)
		{
			auto new_frame = std::make_shared<Audio_frame>();
			Frame_data* frame = new_frame->data();
			frame->sample_rate = sample_rate;
			frame->ch_layout.nb_channels = channel_count;
			frame->nb_samples = static_cast<int>(n_frames);
			frame->format = AV_SAMPLE_FMT_FLT;
			frame->time_base = {1, 1000000};
			frame->pts = static_cast<int64_t>(time_us);
			frame_get_buffer(frame, 32);
			std::memcpy(frame->data[0], samples, n_frames * channel_count * sizeof(float));
			return new_frame;
		}

		// frames -> device interleaved f32, one after the other (extract_samples_interleaved, :150-232).  The frames of a batch are
		// copied into page-locked staging on the CPU and go up as ONE asynchronous copy; on the device
		//   * packed float frames (and mono planar ones) already ARE the interleaved signal: no kernel at all;
		//   * a run of planar stereo float frames of equal length is interleaved by ONE strided launch (frames as "streams");
		//   * integer formats: nae_to_f32_interleaved per frame (the reference's literal divisors).
		// Everything is queued on the stream.  All frames have the channel count of the first.
		float* upload_as_f32(const std::vector<std::shared_ptr<const Audio_frame>>& frames, gpu::Pinned_buffer& h_raw, gpu::Device_buffer& d_raw,
							 gpu::Device_buffer& d_f32, size_t* total_samples)
		{
			nae_ctx* ctx = gpu::context();
			const int ch = frames.front()->data()->ch_layout.nb_channels;
			struct Place { size_t raw_off, stride, plane_bytes, out_off; int planes; bool wire; };
			std::vector<Place> place;
			size_t raw_bytes = 0, out_samples = 0;
			bool all_wire = true;
			for (const auto& f : frames)
			{
				const Frame_data* frame = f->data();
				const int bps = bytes_per_sample(frame->format);
				if (bps == 0 || frame->format == AV_SAMPLE_FMT_DBL)
					throw infra::Processor::Runtime_error(
						"Unsupported sample format", "The processors do not support the given sample format.",
						infra::fmt("Sample format: %d", frame->format)
					);
				const bool planar = sample_fmt_is_planar(frame->format);
				Place p;
				p.planes = planar ? ch : 1;
				p.plane_bytes = (size_t)frame->nb_samples * bps * (planar ? 1 : ch);
				p.wire = frame->format == AV_SAMPLE_FMT_FLT || (frame->format == AV_SAMPLE_FMT_FLTP && ch == 1);
				all_wire = all_wire && p.wire;
				// float planes lie back to back ([frame][ch][n]: what the strided interleave launch reads); integer planes start on 256 bytes
				const bool f32 = frame->format == AV_SAMPLE_FMT_FLT || frame->format == AV_SAMPLE_FMT_FLTP;
				p.stride = f32 ? p.plane_bytes : (p.plane_bytes + 255) / 256 * 256;
				raw_bytes = f32 ? (raw_bytes + 15) / 16 * 16 : (raw_bytes + 255) / 256 * 256;
				p.raw_off = raw_bytes;
				p.out_off = out_samples * ch;
				raw_bytes += p.stride * p.planes;
				out_samples += frame->nb_samples;
				place.push_back(p);
			}
			*total_samples = out_samples;
			auto* host = static_cast<uint8_t*>(h_raw.reserve(raw_bytes));
			float* out = static_cast<float*>(d_f32.reserve(out_samples * ch * sizeof(float)));
			if (all_wire)
			{
				// the staged bytes are the interleaved signal (offsets re-packed without the 16-byte rounding)
				size_t off = 0;
				for (size_t k = 0; k < frames.size(); k++)
				{
					std::memcpy(host + off, frames[k]->data()->data[0], place[k].plane_bytes);
					off += place[k].plane_bytes;
				}
				gpu::check(nae_memcpy_h2d(ctx, out, host, off), "h2d");
				return out;
			}
			auto* raw = static_cast<uint8_t*>(d_raw.reserve(raw_bytes));
			for (size_t k = 0; k < frames.size(); k++)
				for (int q = 0; q < place[k].planes; q++)
					std::memcpy(host + place[k].raw_off + q * place[k].stride, frames[k]->data()->data[q], place[k].plane_bytes);
			gpu::check(nae_memcpy_h2d(ctx, raw, host, raw_bytes), "h2d");
			for (size_t k = 0; k < frames.size();)
			{
				const Frame_data* frame = frames[k]->data();
				const Place& p = place[k];
				if (p.wire)
				{
					gpu::check(nae_memcpy_d2d(ctx, out + p.out_off, raw + p.raw_off, p.plane_bytes), "d2d");
					k++;
				}
				else if (frame->format == AV_SAMPLE_FMT_FLTP)
				{
					// run of planar stereo frames of this length, staged at a constant pitch
					const size_t n = (size_t)frame->nb_samples;
					size_t run = 1;
					while (k + run < frames.size() && frames[k + run]->data()->format == AV_SAMPLE_FMT_FLTP &&
						   (size_t)frames[k + run]->data()->nb_samples == n && place[k + run].raw_off == p.raw_off + run * (place[k + 1].raw_off - p.raw_off))
						run++;
					const size_t pitch_floats = run > 1 ? (place[k + 1].raw_off - p.raw_off) / sizeof(float) : 2 * n;
					const nae_sig src{raw + p.raw_off, pitch_floats, n, 1};
					const nae_sig dst{out + p.out_off, 2 * n, 1, 2};
					gpu::check(nae_copy_sig_f32(ctx, &src, &dst, n, 2, run), "nae_copy_sig_f32");
					k += run;
				}
				else
				{
					const void* pl[2] = {raw + p.raw_off, raw + p.raw_off + p.stride};
					gpu::check(nae_to_f32_interleaved(ctx, frame->format, pl, frame->nb_samples, ch, out + p.out_off), "nae_to_f32_interleaved");
					k++;
				}
			}
			return out;
		}
		// the object soundtouch_process_payload talks to: the phase-vocoder handle (default) or the
		// SoundTouch-shaped WSOLA chain, chosen by the node's "algorithm" key
		struct Stretcher
		{
			nae_stretch* pv = nullptr;
			nae_wsola* st = nullptr;
			~Stretcher()
			{
				if (pv) nae_stretch_destroy(pv);
				if (st) nae_wsola_destroy(st);
			}
			bool open() const { return pv != nullptr || st != nullptr; }
			void create(Stretch_algorithm algo, int sample_rate, int channels, float velocity, float pitch)
			{
				if (algo == Stretch_algorithm::Soundtouch)
					gpu::check(nae_wsola_create(gpu::context(), sample_rate, channels, velocity, pitch, &st), "nae_wsola_create");
				else
					gpu::check(nae_stretch_create(gpu::context(), sample_rate, channels, velocity, pitch, &pv), "nae_stretch_create");
			}
			size_t available() const { return pv ? nae_stretch_available(pv) : nae_wsola_available(st); }
			void put(const float* samples, size_t n)
			{
				gpu::check(pv ? nae_stretch_put(pv, samples, n) : nae_wsola_put(st, samples, n), "stretch put");
			}
			void receive_device(float* dst, size_t max, size_t* got)
			{
				gpu::check(pv ? nae_stretch_receive(pv, dst, max, got) : nae_wsola_receive(st, dst, max, got), "stretch receive");
			}
			void flush() { gpu::check(pv ? nae_stretch_flush(pv) : nae_wsola_flush(st), "stretch flush"); }
		};

		// soundtouch_process_payload, audio-velocity.cpp:265-443, with a GPU handle in SoundTouch's place
		void stretch_process_payload(
			const std::map<std::string, std::shared_ptr<infra::Processor::Product>>& input,
			const std::map<std::string, std::set<std::shared_ptr<infra::Processor::Product>>>& output,
			const std::atomic<bool>& stop_token, float velocity, float pitch, const std::string& processor_name,
			Stretch_algorithm algorithm, Batch_stats& batch_stats
		)
		{
			gpu::Node node;  // this node's context (own stream; device: gpu::pick_device): first local, destroyed last
			batch_stats = {};
			const auto input_item = infra::get_input_item<Audio_stream>(input, "input");
			const auto output_stream = infra::get_output_item<Audio_stream>(output, "output");
			if (!input_item.has_value())
				throw infra::Processor::Runtime_error(
					processor_name + " has no input",
					processor_name + " requires an audio stream input to function properly.",
					"Input item 'input' not found"
				);
			Audio_stream& input_stream = input_item.value().get();
			Stretcher soundtouch;
			gpu::Device_buffer d_raw, d_f32, d_out;
			gpu::Pinned_buffer h_raw, h_out;
			bool input_stream_eof = false, pending_put = false;
			std::shared_ptr<const Audio_frame> held;  // popped, but with another channel count than the batch in front of it
			int channel_count = 0, sample_rate = 0;
			double time_seconds = 0.0;

			// receiveSamples + construct_audio_frame_float + push (:294-316), for every chunk that is ready at once: the chunk sizes are
			// those the reference's loop would take one per turn (min(numSamples, max) while more than `floor` samples are queued); all
			// of them come down as ONE asynchronous copy into page-locked staging behind ONE wait, then they are cut into frames
			const cadence::Bounds chunk_bounds = cadence::bounds(velocity);  // :416-417
			auto acquire_chunks = [&](size_t floor)
			{
				const std::vector<size_t> chunks = cadence::drain(soundtouch.available(), floor, chunk_bounds);
				size_t total = 0;
				for (const size_t take : chunks) total += take;
				if (chunks.empty()) return;
				nae_ctx* ctx = gpu::context();
				float* dev = static_cast<float*>(d_out.reserve(total * channel_count * sizeof(float)));
				float* host = static_cast<float*>(h_out.reserve(total * channel_count * sizeof(float)));
				size_t got = 0;
				soundtouch.receive_device(dev, total, &got);
				gpu::check(nae_memcpy_d2h(ctx, host, dev, got * channel_count * sizeof(float)), "d2h");
				gpu::wait(stop_token);
				size_t pos = 0;
				for (const size_t take : chunks)
				{
					const size_t n = std::min(take, got - pos);
					if (n == 0) break;
					auto new_frame = construct_audio_frame_float(host + pos * channel_count, n, sample_rate, channel_count, (float)(time_seconds * 1000000));
					time_seconds += double(n) / sample_rate;
					pos += n;
					for (auto& stream : output_stream)
						while (!stop_token)
						{
							if (stream->try_push(new_frame) == channel_op_status::success) break;
							nae_fiber::this_fiber::yield();
						}
				}
			};

			while (!stop_token)
			{
				if (!input_stream_eof || held)
				{
					// Batching: the reference puts one frame per round; here every frame that is already waiting
					// (at most 16) is uploaded, converted and put as ONE block behind one wait.  The handle's output does not
					// depend on how its input is cut into puts (tests/test_gpu_stft.py, test_gpu_wsola.py: chunking invariance).
					constexpr size_t max_batch = 16;
					std::vector<std::shared_ptr<const Audio_frame>> batch;
					if (held) batch.push_back(std::move(held));
					held.reset();
					while (batch.size() < max_batch && !input_stream_eof)
					{
						const auto pop_result = input_stream.try_pop();
						if (!pop_result.has_value())
						{
							if (pop_result.error() != channel_op_status::empty)
								throw infra::Processor::Runtime_error(
									"Unexpected error when fetching audio frame", processor_name + " encountered an unexpected error.",
									infra::fmt("Channel fetch error: %d", (int)pop_result.error())
								);
							if (input_stream.eof()) input_stream_eof = true;
							break;
						}
						if (!batch.empty() && pop_result.value()->data()->ch_layout.nb_channels != batch.front()->data()->ch_layout.nb_channels)
						{
							held = pop_result.value();
							break;
						}
						batch.push_back(pop_result.value());
					}
					if (!batch.empty())
					{
						constexpr size_t max_queued_samples = 65536;
						const Frame_data* frame = batch.front()->data();
						if (!soundtouch.open())
						{
							if (frame->sample_rate < 8000 || frame->sample_rate > 48000)  // :371-379
								throw infra::Processor::Runtime_error(
									"Unsupported sample rate",
									infra::fmt("%d requires a sample rate between 8000 and 48000 Hz.", frame->sample_rate),
									infra::fmt("Sample rate: %d", frame->sample_rate)
								);
							soundtouch.create(algorithm, frame->sample_rate, frame->ch_layout.nb_channels, velocity, pitch);
							channel_count = frame->ch_layout.nb_channels;
							time_seconds = frame->pts * av_q2d(frame->time_base);
							sample_rate = frame->sample_rate;
						}
						// (:399-400 waits while more than 65536 samples are queued in SoundTouch.  Only this fiber takes samples out, and a
						// batched put adds up to 16 frames at once, so instead of waiting the loop below receives until the queue is
						// under one chunk again: the queue never grows beyond one batch.)
						static_assert(max_queued_samples >= 16 * 1152 * 3, "a batch fits the reference's queue bound");
						size_t total = 0;
						float* samples = upload_as_f32(batch, h_raw, d_raw, d_f32, &total);
						soundtouch.put(samples, total);
						// (no wait here: the frames were copied into page-locked staging and are released; staging and device buffers are
						// next touched behind the wait of the receive below, or of the next turn's)
						pending_put = true;
						batch_stats.rounds += batch.size();
					}
				}
				if (soundtouch.open())
				{
					// (:414 "numSamples() == 0 && eof -> break" is subsumed by the flush branch)
					const uint32_t min_samples = chunk_bounds.min_samples;
					if (soundtouch.available() > min_samples)
					{
						// the reference receives one chunk per loop turn because it puts one frame per turn (:403,416-424); a batched put
						// makes several chunks available, and all of them are taken now (chunk sizes stay inside [min, max])
						acquire_chunks(min_samples);
						pending_put = false;
						batch_stats.waits++;
					}
					else if (input_stream_eof)
					{
						soundtouch.flush();
						// the reference emits ONE frame with everything that is left (:427-433); here flush() may release
						// the whole stream, so it is cut into the same [min, max] chunks the steady state uses
						acquire_chunks(0);
						break;
					}
					if (pending_put)
					{
						// a put that released nothing: its staging is reused by the next batch, so it is waited for now
						gpu::wait(stop_token);
						pending_put = false;
						batch_stats.waits++;
					}
				}
				else if (input_stream_eof)
					break;
				nae_fiber::this_fiber::yield();
			}
			for (auto& stream : output_stream) stream->set_eof();
		}
	}

	// ------------------------------------------------------------------------------------------ Velocity_modifier
	infra::Processor::Info Velocity_modifier::get_processor_info()
	{
		return {"velocity_modifier", "Velocity Modifier", false,
				[] { return std::unique_ptr<infra::Processor>(new Velocity_modifier); }, "Audio Velocity Modifier (MI355X)"};
	}
	std::vector<infra::Processor::Pin_attribute> Velocity_modifier::get_pin_attributes() const { return io_pins(); }

	void Velocity_modifier::process_payload(
		const std::map<std::string, std::shared_ptr<infra::Processor::Product>>& input,
		const std::map<std::string, std::set<std::shared_ptr<infra::Processor::Product>>>& output,
		const std::atomic<bool>& stop_token, std::any&
	)
	{
		stretch_process_payload(input, output, stop_token, velocity, keep_pitch ? 1 / velocity : 1, get_processor_info().display_name,
								algorithm, batch_stats);  // :452-459
	}

	Json::Value Velocity_modifier::serialize() const
	{
		Json::Value value;
		value["velocity"] = velocity;
		value["keep_pitch"] = keep_pitch;
		if (algorithm != default_stretch_algorithm()) value["algorithm"] = algorithm_name(algorithm);
		return value;
	}

	void Velocity_modifier::deserialize(const Json::Value& value)
	{
		if (value.isMember("velocity") && value["velocity"].isDouble()) velocity = value["velocity"].asFloat();
		if (value.isMember("keep_pitch") && value["keep_pitch"].isBool()) keep_pitch = value["keep_pitch"].asBool();
		algorithm = algorithm_from_json(value);
	}

	// ------------------------------------------------------------------------------------------ Pitch_modifier
	infra::Processor::Info Pitch_modifier::get_processor_info()
	{
		return {"pitch_modifier", "Pitch Modifier", false, [] { return std::unique_ptr<infra::Processor>(new Pitch_modifier); },
				"Audio Pitch Modifier (MI355X)"};
	}
	std::vector<infra::Processor::Pin_attribute> Pitch_modifier::get_pin_attributes() const { return io_pins(); }

	void Pitch_modifier::process_payload(
		const std::map<std::string, std::shared_ptr<infra::Processor::Product>>& input,
		const std::map<std::string, std::set<std::shared_ptr<infra::Processor::Product>>>& output,
		const std::atomic<bool>& stop_token, std::any&
	)
	{
		stretch_process_payload(input, output, stop_token, 1, std::pow(2.0f, pitch / 12.0f), get_processor_info().display_name,
								algorithm, batch_stats);  // :469-476
	}

	Json::Value Pitch_modifier::serialize() const
	{
		Json::Value value;
		value["pitch"] = pitch;
		if (algorithm != default_stretch_algorithm()) value["algorithm"] = algorithm_name(algorithm);
		return value;
	}
	void Pitch_modifier::deserialize(const Json::Value& value)
	{
		if (value.isMember("pitch") && value["pitch"].isDouble()) pitch = value["pitch"].asFloat();
		algorithm = algorithm_from_json(value);
	}

	// ------------------------------------------------------------------------------------------ Audio_spectrum
	infra::Processor::Info Audio_spectrum::get_processor_info()
	{
		return {"audio_spectrum", "FFT Spectrum", false, [] { return std::unique_ptr<infra::Processor>(new Audio_spectrum); },
				"Hann-windowed 1024-point magnitude spectrum every 256 samples (MI355X)"};
	}
	std::vector<infra::Processor::Pin_attribute> Audio_spectrum::get_pin_attributes() const { return io_pins(); }

	void Audio_spectrum::process_payload(
		const std::map<std::string, std::shared_ptr<infra::Processor::Product>>& input,
		const std::map<std::string, std::set<std::shared_ptr<infra::Processor::Product>>>& output,
		const std::atomic<bool>& stop_token, std::any&
	)
	{
		gpu::Node node;  // this node's context (own stream): first local, destroyed last — before the handle guard and the buffers
		const auto input_item = infra::get_input_item<Audio_stream>(input, "input");
		const auto output_stream = infra::get_output_item<Audio_stream>(output, "output");
		if (!input_item.has_value())
			throw Runtime_error("FFT Spectrum has no input", "FFT Spectrum requires an audio stream input to function properly.", "Input item 'input' not found");
		Audio_stream& input_stream = input_item.value().get();
		nae_ctx* ctx = gpu::context();
		last_context = ctx;
		nae_spectrum* spectrum = nullptr;
		struct Guard { nae_spectrum*& h; ~Guard() { if (h) nae_spectrum_destroy(h); } } guard{spectrum};
		gpu::Device_buffer d_raw, d_f32, d_out;
		gpu::Pinned_buffer h_raw, h_out;
		int ch = 0, sample_rate = 0;
		double time_seconds = 0.0;
		std::shared_ptr<const Audio_frame> held;  // popped, but with another channel count than the batch in front of it

		while (!stop_token)
		{
			// every frame that is already waiting (at most 16) is uploaded and put as one block behind one wait (the handle's
			// frames do not depend on how its input is cut into puts)
			constexpr size_t max_batch = 16;
			std::vector<std::shared_ptr<const Audio_frame>> batch;
			if (held) batch.push_back(std::move(held));
			held.reset();
			bool ended = false;
			while (batch.size() < max_batch)
			{
				const auto pop_result = input_stream.try_pop();
				if (!pop_result.has_value())
				{
					ended = input_stream.eof();
					break;
				}
				if (!batch.empty() && pop_result.value()->data()->ch_layout.nb_channels != batch.front()->data()->ch_layout.nb_channels)
				{
					held = pop_result.value();
					break;
				}
				batch.push_back(pop_result.value());
			}
			if (batch.empty())
			{
				if (ended) break;  // a trailing partial window produces no frame
				nae_fiber::this_fiber::yield();
				continue;
			}
			const Frame_data* frame = batch.front()->data();
			if (spectrum == nullptr)
			{
				ch = frame->ch_layout.nb_channels;
				if (ch != 1 && ch != 2) throw Runtime_error("Invalid channel count", "Only mono and stereo audio are supported.", infra::fmt("Got %d channels", ch));
				sample_rate = frame->sample_rate;
				time_seconds = frame->pts * av_q2d(frame->time_base);
				gpu::check(nae_spectrum_create(ctx, 1024, 256, ch, &spectrum), "nae_spectrum_create");
			}
			size_t total = 0;
			float* samples = upload_as_f32(batch, h_raw, d_raw, d_f32, &total);
			gpu::check(nae_spectrum_put(spectrum, samples, total), "nae_spectrum_put");
			const size_t ready = nae_spectrum_available(spectrum);
			if (ready == 0) { gpu::wait(stop_token); continue; }
			const size_t rec = (size_t)ch * 513;
			float* dout = static_cast<float*>(d_out.reserve(ready * rec * sizeof(float)));
			float* host = static_cast<float*>(h_out.reserve(ready * rec * sizeof(float)));
			size_t got = 0;
			gpu::check(nae_spectrum_receive(spectrum, dout, ready, &got), "nae_spectrum_receive");
			gpu::check(nae_memcpy_d2h(ctx, host, dout, got * rec * sizeof(float)), "d2h");
			gpu::wait(stop_token);
			for (size_t f = 0; f < got && !stop_token; f++)
			{
				auto out = std::make_shared<Audio_frame>();
				Frame_data* o = out->data();
				o->format = AV_SAMPLE_FMT_FLTP;
				o->sample_rate = sample_rate;
				o->nb_samples = 513;
				o->ch_layout.nb_channels = ch;
				o->time_base = {1, 1000000};
				o->pts = (int64_t)(time_seconds * 1000000);
				frame_get_buffer(o, 32);
				for (int c = 0; c < ch; c++) std::memcpy(o->data[c], host + (f * ch + c) * 513, 513 * sizeof(float));
				time_seconds += 256.0 / sample_rate;
				for (auto& stream : output_stream)
					while (!stop_token && stream->try_push(out) != channel_op_status::success) nae_fiber::this_fiber::yield();
			}
		}
		for (auto& stream : output_stream) stream->set_eof();
	}
}
