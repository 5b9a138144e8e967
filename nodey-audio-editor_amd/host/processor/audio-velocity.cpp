#include "audio-velocity.hpp"
#include "gpu-context.hpp"

#include <algorithm>
#include <cmath>

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
		std::shared_ptr<Audio_frame> construct_audio_frame_float(const std::vector<float>& samples, int sample_rate,
																 int channel_count, float time_us)
		{
			auto new_frame = std::make_shared<Audio_frame>();
			Frame_data* frame = new_frame->data();
			frame->sample_rate = sample_rate;
			frame->ch_layout.nb_channels = channel_count;
			frame->nb_samples = static_cast<int>(samples.size() / channel_count);
			frame->format = AV_SAMPLE_FMT_FLT;
			frame->time_base = {1, 1000000};
			frame->pts = static_cast<int64_t>(time_us);
			frame_get_buffer(frame, 32);
			std::copy(samples.begin(), samples.end(), reinterpret_cast<float*>(frame->data[0]));
			return new_frame;
		}

		// frames -> device interleaved f32, one after the other (extract_samples_interleaved, :150-232, as
		// nae_to_f32_interleaved per frame).  Everything is queued on the stream; the frames must stay alive until the caller
		// has waited for it.  All frames have the channel count of the first.
		float* upload_as_f32(const std::vector<std::shared_ptr<const Audio_frame>>& frames, gpu::Device_buffer& d_raw,
							 gpu::Device_buffer& d_f32, size_t* total_samples)
		{
			nae_ctx* ctx = gpu::context();
			const int ch = frames.front()->data()->ch_layout.nb_channels;
			struct Place { size_t raw_off, stride, plane_bytes, out_off; int planes; };
			std::vector<Place> place;
			size_t raw_bytes = 0, out_samples = 0;
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
				p.stride = (p.plane_bytes + 255) / 256 * 256;
				p.raw_off = raw_bytes;
				p.out_off = out_samples * ch;
				raw_bytes += p.stride * p.planes;
				out_samples += frame->nb_samples;
				place.push_back(p);
			}
			auto* raw = static_cast<uint8_t*>(d_raw.reserve(raw_bytes));
			float* out = static_cast<float*>(d_f32.reserve(out_samples * ch * sizeof(float)));
			for (size_t k = 0; k < frames.size(); k++)
			{
				const Frame_data* frame = frames[k]->data();
				const Place& p = place[k];
				const void* pl[2] = {raw + p.raw_off, raw + p.raw_off + p.stride};
				for (int q = 0; q < p.planes; q++)
					gpu::check(nae_memcpy_h2d(ctx, raw + p.raw_off + q * p.stride, frame->data[q], p.plane_bytes), "h2d");
				gpu::check(nae_to_f32_interleaved(ctx, frame->format, pl, frame->nb_samples, ch, out + p.out_off), "nae_to_f32_interleaved");
			}
			*total_samples = out_samples;
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
			void receive_host(float* dst, size_t max, size_t* got)
			{
				gpu::check(pv ? nae_stretch_receive_host(pv, dst, max, got) : nae_wsola_receive_host(st, dst, max, got), "stretch receive");
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
			gpu::Node node;  // this node's context (own stream, device by round-robin): first local, destroyed last
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
			gpu::Device_buffer d_raw, d_f32;
			bool input_stream_eof = false;
			std::shared_ptr<const Audio_frame> held;  // popped, but with another channel count than the batch in front of it
			const double time_ratio = 1.0f / velocity;
			int channel_count = 0, sample_rate = 0;
			double time_seconds = 0.0;

			auto acquire_func = [&](int count)
			{
				std::vector<float> output_samples((size_t)count * channel_count);
				size_t samples_read = 0;
				gpu::wait(stop_token);
				soundtouch.receive_host(output_samples.data(), count, &samples_read);
				output_samples.resize(samples_read * channel_count);
				auto new_frame = construct_audio_frame_float(output_samples, sample_rate, channel_count, (float)(time_seconds * 1000000));
				time_seconds += double(samples_read) / sample_rate;
				for (auto& stream : output_stream)
					while (!stop_token)
					{
						if (stream->try_push(new_frame) == channel_op_status::success) break;
						nae_fiber::this_fiber::yield();
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
						float* samples = upload_as_f32(batch, d_raw, d_f32, &total);
						soundtouch.put(samples, total);
						gpu::wait(stop_token);  // d_raw / d_f32 are reused by the next batch; the frames are released
						batch_stats.rounds += batch.size();
						batch_stats.waits++;
					}
				}
				if (soundtouch.open())
				{
					// (:414 "numSamples() == 0 && eof -> break" is subsumed by the flush branch)
					const uint32_t min_samples = time_ratio * 1152;
					const uint32_t max_samples = time_ratio * 1152 * 3;
					if (soundtouch.available() > min_samples)
					{
						// the reference receives one chunk per loop turn because it puts one frame per turn (:403,416-424); a batched put
						// makes several chunks available, and all of them are taken now (chunk sizes stay inside [min, max])
						while (!stop_token && soundtouch.available() > min_samples)
							acquire_func((int)std::min<size_t>(soundtouch.available(), max_samples));
					}
					else if (input_stream_eof)
					{
						soundtouch.flush();
						// the reference emits ONE frame with everything that is left (:427-433); here flush() may release
						// the whole stream, so it is cut into the same [min, max] chunks the steady state uses
						while (!stop_token && soundtouch.available() > 0)
							acquire_func((int)std::min<size_t>(soundtouch.available(), std::max<uint32_t>(max_samples, 1)));
						break;
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
		const auto input_item = infra::get_input_item<Audio_stream>(input, "input");
		const auto output_stream = infra::get_output_item<Audio_stream>(output, "output");
		if (!input_item.has_value())
			throw Runtime_error("FFT Spectrum has no input", "FFT Spectrum requires an audio stream input to function properly.", "Input item 'input' not found");
		Audio_stream& input_stream = input_item.value().get();
		nae_ctx* ctx = gpu::context();
		nae_spectrum* spectrum = nullptr;
		struct Guard { nae_spectrum*& h; ~Guard() { if (h) nae_spectrum_destroy(h); } } guard{spectrum};
		gpu::Device_buffer d_raw, d_f32, d_out;
		int ch = 0, sample_rate = 0;
		double time_seconds = 0.0;
		std::vector<float> host;

		while (!stop_token)
		{
			const auto pop_result = input_stream.try_pop();
			if (!pop_result.has_value())
			{
				if (input_stream.eof()) break;  // a trailing partial window produces no frame
				nae_fiber::this_fiber::yield();
				continue;
			}
			const Frame_data* frame = pop_result.value()->data();
			if (spectrum == nullptr)
			{
				ch = frame->ch_layout.nb_channels;
				if (ch != 1 && ch != 2) throw Runtime_error("Invalid channel count", "Only mono and stereo audio are supported.", infra::fmt("Got %d channels", ch));
				sample_rate = frame->sample_rate;
				time_seconds = frame->pts * av_q2d(frame->time_base);
				gpu::check(nae_spectrum_create(ctx, 1024, 256, ch, &spectrum), "nae_spectrum_create");
			}
			size_t total = 0;
			float* samples = upload_as_f32({pop_result.value()}, d_raw, d_f32, &total);
			gpu::check(nae_spectrum_put(spectrum, samples, total), "nae_spectrum_put");
			const size_t ready = nae_spectrum_available(spectrum);
			if (ready == 0) { gpu::wait(stop_token); continue; }
			const size_t rec = (size_t)ch * 513;
			float* dout = static_cast<float*>(d_out.reserve(ready * rec * sizeof(float)));
			size_t got = 0;
			gpu::check(nae_spectrum_receive(spectrum, dout, ready, &got), "nae_spectrum_receive");
			host.resize(got * rec);
			gpu::check(nae_memcpy_d2h(ctx, host.data(), dout, host.size() * sizeof(float)), "d2h");
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
				for (int c = 0; c < ch; c++)
					std::copy(host.begin() + (f * ch + c) * 513, host.begin() + (f * ch + c + 1) * 513, reinterpret_cast<float*>(o->data[c]));
				time_seconds += 256.0 / sample_rate;
				for (auto& stream : output_stream)
					while (!stop_token && stream->try_push(out) != channel_op_status::success) nae_fiber::this_fiber::yield();
			}
		}
		for (auto& stream : output_stream) stream->set_eof();
	}
}
