// selftest.cpp — drives the host mirror (nodey-audio-editor_amd/host) the way the reference's Runner drives its
// processors: one fiber per node, bounded Audio_streams between them.  `selftest cpu` checks the scheduler / stream /
// registry / error plumbing without a GPU; `selftest gpu` runs real graphs through the GPU processors and compares
// with the CPU oracle (this file is test code: linking the oracle here is allowed).
#include "infra/runner.hpp"
#include "processor/audio-mix.hpp"
#include "processor/audio-velocity.hpp"
#include "processor/gpu-context.hpp"
#include "processor/audio-vol.hpp"
#include "../../oracle/nae_oracle.h"

#include "processor/bimix-align.hpp"
#include "processor/velocity-cadence.hpp"
#include <cmath>
#include <cstring>
#include <iostream>

using namespace processor;
using infra::Runner;

static int failures = 0;
#define CHECK(cond, msg)                                                          \
	do {                                                                          \
		if (!(cond)) { std::cout << "FAIL " << __LINE__ << ": " << msg << "\n"; failures++; } \
	} while (0)

// ---- test-only nodes
class Test_source : public infra::Processor
{
  public:

	std::vector<float> samples;  // interleaved
	int ch = 2, frame_size = 1152, format = AV_SAMPLE_FMT_FLT, sample_rate = 48000;
	double start_seconds = 0.0;
	bool one_at_a_time = false;  // push the next frame only once every consumer has taken the previous one (frames arrive singly)

	static Info get_processor_info() { return {"test_source", "Test Source", false, [] { return std::unique_ptr<Processor>(new Test_source); }, ""}; }
	Info get_processor_info_non_static() const override { return get_processor_info(); }
	std::vector<Pin_attribute> get_pin_attributes() const override
	{
		return {{"output", "Output", typeid(Audio_stream), false, [] { return std::make_shared<Audio_stream>(); }}};
	}
	Json::Value serialize() const override { return {}; }
	void deserialize(const Json::Value&) override {}
	void draw_title() override {}
	bool draw_content(bool) override { return false; }
	void process_payload(const std::map<std::string, std::shared_ptr<Product>>&,
						 const std::map<std::string, std::set<std::shared_ptr<Product>>>& output,
						 const std::atomic<bool>& stop_token, std::any&) override
	{
		const auto outs = infra::get_output_item<Audio_stream>(output, "output");
		const size_t total = samples.size() / ch;
		for (size_t pos = 0; pos < total && !stop_token; pos += frame_size)
		{
			const int n = (int)std::min<size_t>(frame_size, total - pos);
			auto frame = std::make_shared<Audio_frame>();
			Frame_data* f = frame->data();
			f->format = format;
			f->sample_rate = sample_rate;
			f->nb_samples = n;
			f->ch_layout.nb_channels = ch;
			f->time_base = {1, 1000000};
			f->pts = (int64_t)((start_seconds + double(pos) / sample_rate) * 1000000);
			frame_get_buffer(f, 32);
			const float* src = samples.data() + pos * ch;
			if (format == AV_SAMPLE_FMT_FLT) std::memcpy(f->data[0], src, (size_t)n * ch * sizeof(float));
			else if (format == AV_SAMPLE_FMT_FLTP)
				for (int c = 0; c < ch; c++)
					for (int i = 0; i < n; i++) reinterpret_cast<float*>(f->data[c])[i] = src[i * ch + c];
			else if (format == AV_SAMPLE_FMT_S16)
				for (int i = 0; i < n * ch; i++) reinterpret_cast<int16_t*>(f->data[0])[i] = (int16_t)std::lrintf(src[i] * 32767.0f);
			for (auto& o : outs)
				while (!stop_token && o->try_push(frame) != channel_op_status::success) nae_fiber::this_fiber::yield();
			if (one_at_a_time)
				for (auto& o : outs)
					while (!stop_token && o->buffered_count() > 0) nae_fiber::this_fiber::yield();
		}
		for (auto& o : outs) o->set_eof();
	}
};

class Test_sink : public infra::Processor
{
  public:

	std::vector<std::shared_ptr<const Audio_frame>> frames;
	size_t max_fill = 0;
	bool lazy_consumer = true;   // pops on every 4th turn only: forces back-pressure (the tests); false: the benchmark's sink
	bool keep = true;            // false: count the frames, drop the samples
	size_t n_frames = 0, n_samples = 0;

	static Info get_processor_info() { return {"test_sink", "Test Sink", false, [] { return std::unique_ptr<Processor>(new Test_sink); }, ""}; }
	Info get_processor_info_non_static() const override { return get_processor_info(); }
	std::vector<Pin_attribute> get_pin_attributes() const override
	{
		return {{"input", "Input", typeid(Audio_stream), true, [] { return std::make_shared<Audio_stream>(); }}};
	}
	Json::Value serialize() const override { return {}; }
	void deserialize(const Json::Value&) override {}
	void draw_title() override {}
	bool draw_content(bool) override { return false; }
	void process_payload(const std::map<std::string, std::shared_ptr<Product>>& input,
						 const std::map<std::string, std::set<std::shared_ptr<Product>>>&, const std::atomic<bool>& stop_token,
						 std::any&) override
	{
		auto in = infra::get_input_item<Audio_stream>(input, "input");
		if (!in.has_value()) throw Runtime_error("sink has no input", "", "");
		Audio_stream& s = in.value().get();
		int lazy = 0;
		while (!stop_token)
		{
			max_fill = std::max(max_fill, s.buffered_count());
			if (lazy_consumer && (++lazy & 3) != 0) { nae_fiber::this_fiber::yield(); continue; }  // a slow consumer: forces back-pressure
			auto r = s.try_pop();
			if (!r.has_value())
			{
				if (s.eof()) break;
				nae_fiber::this_fiber::yield();
				continue;
			}
			n_frames++;
			n_samples += (size_t)r.value()->data()->nb_samples;
			if (keep) frames.push_back(r.value());
		}
	}
	std::vector<float> interleaved() const  // FLT frames concatenated
	{
		std::vector<float> out;
		for (auto& f : frames)
		{
			const Frame_data* d = f->data();
			const float* p = reinterpret_cast<const float*>(d->data[0]);
			out.insert(out.end(), p, p + (size_t)d->nb_samples * d->ch_layout.nb_channels);
		}
		return out;
	}
};

static std::vector<float> uniform(size_t n, uint64_t seed)
{
	std::vector<float> v(n);
	orc_fill_uniform(v.data(), n, seed);
	return v;
}

static double rel_rms(const std::vector<float>& a, const std::vector<float>& b)
{
	if (a.size() != b.size()) return 1e9;
	double e = 0, r = 0;
	for (size_t i = 0; i < a.size(); i++) { e += (double(a[i]) - b[i]) * (double(a[i]) - b[i]); r += double(b[i]) * b[i]; }
	return std::sqrt(e / std::max(r, 1e-300));
}

// ---- the velocity / pitch node's output cadence (/root/reference/src/processor/audio-velocity.cpp:399-436), restated as the
// reference runs it: per turn at most ONE put of one source frame and at most ONE receive.  `frames_per_turn` > 1 restates the
// mirror's batched turn instead (that many frames put as one block, then the receive rule until it no longer fires).
struct Cadence { std::vector<size_t> sizes; std::vector<int64_t> pts; size_t total = 0; };

template <typename Put, typename Avail, typename Receive, typename Flush>
static Cadence run_cadence(size_t S, int frame_size, float velocity, bool mirror, size_t frames_per_turn, Put put, Avail avail, Receive receive, Flush flush)
{
	const processor::cadence::Bounds b = processor::cadence::bounds(velocity);
	Cadence c;
	double time_seconds = 0.0;
	auto emit = [&](size_t n) {
		const size_t got = receive(n);
		c.sizes.push_back(got);
		c.pts.push_back((int64_t)(float)(time_seconds * 1000000));   // construct_audio_frame_float's float-typed clock (:238,249)
		time_seconds += double(got) / 48000;
		c.total += got;
	};
	size_t pos = 0;
	bool eof = false;
	for (;;)
	{
		if (!eof)
		{
			for (size_t k = 0; k < frames_per_turn && pos < S; k++)
			{
				const size_t n = std::min<size_t>(frame_size, S - pos);
				put(pos, n);
				pos += n;
			}
			// (the reference learns about the end of its input on the turn AFTER the last frame: try_pop fails and eof() is set)
		}
		if (mirror)
		{
			if (avail() > b.min_samples)
				for (const size_t n : processor::cadence::drain(avail(), b.min_samples, b)) emit(n);
			else if (eof)
			{
				flush();
				for (const size_t n : processor::cadence::drain(avail(), 0, b)) emit(n);
				break;
			}
		}
		else
		{
			if (const size_t n = processor::cadence::reference_receive(avail(), b)) emit(n);
			else if (eof)
			{
				flush();
				if (avail() > 0) emit(avail());                      // ONE frame with everything that is left (:427-433)
				break;
			}
		}
		if (pos >= S && !eof && avail() <= b.min_samples) eof = true;   // an empty turn: the source has ended
		else if (pos >= S && !eof && !mirror && processor::cadence::reference_receive(avail(), b) == 0) eof = true;
	}
	return c;
}

static Cadence oracle_cadence(const std::vector<float>& x, int frame_size, double rate, double pitch, float velocity, bool mirror, size_t frames_per_turn)
{
	orc_st* st = nullptr;
	orc_st_create(48000, 2, rate, pitch, &st);
	std::vector<float> sink;
	Cadence c = run_cadence(
		x.size() / 2, frame_size, velocity, mirror, frames_per_turn, [&](size_t pos, size_t n) { orc_st_put(st, x.data() + 2 * pos, n); },
		[&] { return orc_st_available(st); },
		[&](size_t n) { sink.resize(2 * n); return orc_st_receive(st, sink.data(), n); }, [&] { orc_st_flush(st); });
	orc_st_destroy(st);
	return c;
}

// ------------------------------------------------------------------------------------------------ CPU-only checks
static void test_streams_and_scheduler()
{
	Audio_stream s;
	auto f = std::make_shared<Audio_frame>();
	for (int i = 0; i < 16; i++) CHECK(s.try_push(f) == channel_op_status::success, "push " << i);
	CHECK(s.try_push(f) == channel_op_status::full, "17th push must report full (capacity 16, config.hpp:53)");
	CHECK(s.buffered_count() == 16, "fill counter");
	for (int i = 0; i < 16; i++) CHECK(s.try_pop().has_value(), "pop " << i);
	auto e = s.try_pop();
	CHECK(!e.has_value() && e.error() == channel_op_status::empty, "empty status");
	CHECK(!s.eof(), "eof flag is separate from the queue");
	s.set_eof();
	CHECK(s.eof(), "set_eof");

	// source -> sink, 200 frames through a 16-deep stream with a slow consumer
	Runner r;
	auto src = std::make_shared<Test_source>();
	src->samples = uniform(200 * 64 * 2, 1);
	src->frame_size = 64;
	auto sink = std::make_shared<Test_sink>();
	r.add_node(1, src);
	r.add_node(2, sink);
	r.add_link({1, "output", 2, "input"});
	CHECK(r.run(), "runner ok");
	CHECK(sink->frames.size() == 200, "all frames arrive: " << sink->frames.size());
	CHECK(sink->interleaved() == src->samples, "payload intact and in order");
	CHECK(sink->max_fill == 16, "producer ran into back-pressure (max fill " << sink->max_fill << ")");
	CHECK(r.context_switches() > 400, "fibers interleave");
	CHECK(sink->frames[3]->data()->pts == (int64_t)(3 * 64 / 48000.0 * 1000000), "pts");
	for (auto& [id, res] : r.get_processor_resources()) CHECK(res->state == Runner::State::Finished, "state of node " << id);
}

static void test_registry_and_json()
{
	infra::Processor::processor_map.clear();
	infra::register_all_processors();
	const char* ids[] = {"audio_volume_adjust", "velocity_modifier", "pitch_modifier", "audio_amix", "audio_bimix", "audio_bimix_v2", "audio_spectrum"};
	for (auto id : ids) CHECK(infra::Processor::processor_map.count(id) == 1, "registered " << id);
	bool threw = false;
	try { infra::register_all_processors(); } catch (const std::logic_error&) { threw = true; }
	CHECK(threw, "duplicate registration throws logic_error (processor.hpp:122-126)");
	auto amix = infra::Processor::processor_map["audio_amix"].generate();
	CHECK(amix->get_pin_attributes().size() == 3, "amix default pins: output + input_1 + input_2");
	Json::Value v;
	v["input_num"] = 3;
	for (int i = 0; i < 3; i++) { v[infra::fmt("volumes%d", i)] = 0.25 * (i + 1); v[infra::fmt("locks%d", i)] = (i == 1); }
	amix->deserialize(v);
	CHECK(amix->get_pin_attributes().size() == 4 && amix->get_pin_attributes()[3].identifier == "input_3", "pins follow input_num");
	auto back = amix->serialize();
	CHECK(back["input_num"].asInt() == 3 && back["volumes2"].asFloat() == 0.75f && back["locks1"].asBool(), "amix JSON round trip");
	bool bad = false;
	try { amix->deserialize(Json::Value()); } catch (const infra::Processor::Runtime_error& e) { bad = e.detail == "Wrong field: input_num"; }
	CHECK(bad, "missing input_num -> Runtime_error");
	auto bimix = infra::Processor::processor_map["audio_bimix"].generate();
	Json::Value b;
	b["bias"] = 7.0;
	bimix->deserialize(b);
	CHECK(bimix->serialize()["bias"].asFloat() == 1.0f, "bias clamps to [-1, 1] (audio-bimix.cpp:381-382)");
	auto vel = infra::Processor::processor_map["velocity_modifier"].generate();
	Json::Value q;
	q["velocity"] = 1.5;
	q["keep_pitch"] = true;
	vel->deserialize(q);
	CHECK(vel->serialize()["velocity"].asFloat() == 1.5f && vel->serialize()["keep_pitch"].asBool(), "velocity JSON");
	CHECK(infra::Processor::processor_map["audio_volume_adjust"].generate()->serialize().isNull(), "volume is not serialised (audio-vol.hpp:57-58)");
}

// what the mirror's registry holds, one line per processor, for tests/test_ref_pin.py to compare with the TEXT of the reference's sources:
// identifier, display name, singleton flag, the pins of a default-constructed node (identifier, direction) and the keys its serialize() emits
static void print_registry()
{
	infra::Processor::processor_map.clear();
	infra::register_all_processors();
	for (const auto& [id, info] : infra::Processor::processor_map)
	{
		auto node = info.generate();          // make_unique<T>: compiles only if T overrides every pure virtual of infra::Processor
		node->draw_title();
		(void)node->draw_content(true);
		std::cout << "PROC {\"identifier\": \"" << info.identifier << "\", \"display_name\": \"" << info.display_name << "\", \"singleton\": "
				  << (info.singleton ? "true" : "false") << ", \"same_info_non_static\": "
				  << (node->get_processor_info_non_static().identifier == info.identifier ? "true" : "false") << ", \"pins\": [";
		bool first = true;
		for (const auto& pin : node->get_pin_attributes())
		{
			std::cout << (first ? "" : ", ") << "[\"" << pin.identifier << "\", " << (pin.is_input ? "true" : "false") << ", "
					  << (pin.type.get() == typeid(Audio_stream) && pin.generate_func && dynamic_cast<Audio_stream*>(pin.generate_func().get()) ? "true" : "false") << "]";
			first = false;
		}
		std::cout << "], \"json_keys\": [";
		first = true;
		const Json::Value v = node->serialize();
		for (const auto& key : v.getMemberNames())
		{
			std::cout << (first ? "" : ", ") << "\"" << key << "\"";
			first = false;
		}
		std::cout << "]}\n";
	}
}

// the GUI hooks without a GUI (draw-headless.cpp): what the reference's bodies do to the parameters when no widget is touched
static void test_headless_draw_hooks()
{
	Audio_vol vol;
	vol.set_volume(3.0f);
	CHECK(!vol.draw_content(false) && vol.get_volume() == 3.0f, "volume inside [0, 10] stays (audio-vol.cpp:266-274)");
	Audio_amix amix;
	Json::Value v;
	v["input_num"] = 3;
	const float w[3] = {0.5f, 0.25f, 0.75f};
	for (int i = 0; i < 3; i++) { v[infra::fmt("volumes%d", i)] = w[i]; v[infra::fmt("locks%d", i)] = (i == 1); }
	amix.deserialize(v);
	CHECK(!amix.draw_content(false), "no pin change reported");
	const Json::Value back = amix.serialize();
	// audio-amix.cpp:379-387: the unlocked weights are divided by their sum (1.25), a locked one stays
	CHECK(back["volumes0"].asFloat() == 0.5f / 1.25f && back["volumes1"].asFloat() == 0.25f && back["volumes2"].asFloat() == 0.75f / 1.25f,
		  "mixer weights renormalised as the GUI does: " << back["volumes0"].asFloat() << " " << back["volumes1"].asFloat() << " " << back["volumes2"].asFloat());
}

static void test_error_capture()
{
	// a node without its input link ends in State::Error with the reference's message, and stops the graph
	Runner r;
	auto vol = std::make_shared<Audio_vol>();
	auto sink = std::make_shared<Test_sink>();
	r.add_node(1, vol);
	r.add_node(2, sink);
	r.add_link({1, "output", 2, "input"});
	CHECK(!r.run(), "runner reports the error");
	auto& res = r.get_processor_resources().at(1);
	CHECK(res->state == Runner::State::Error, "state Error");
	bool ok = false;
	try { std::any_cast<infra::Processor::Runtime_error>(res->exception); ok = true; } catch (...) {}
	CHECK(ok, "exception is a Processor::Runtime_error");
	CHECK(res->error_text.find("Volume adjust processor has no input") == 0, res->error_text);
}

// ------------------------------------------------------------------------------------------------ GPU graphs
static void test_gpu_volume()
{
	for (int format : {AV_SAMPLE_FMT_FLT, AV_SAMPLE_FMT_FLTP, AV_SAMPLE_FMT_S16})
	{
		Runner r;
		auto src = std::make_shared<Test_source>();
		src->samples = uniform(20000 * 2, 3);
		src->format = format;
		auto vol = std::make_shared<Audio_vol>();
		vol->set_volume(0.5f);
		auto sink = std::make_shared<Test_sink>();
		r.add_node(1, src);
		r.add_node(2, vol);
		r.add_node(3, sink);
		r.add_link({1, "output", 2, "input"});
		r.add_link({2, "output", 3, "input"});
		const bool ok = r.run();
		CHECK(ok, "volume graph runs (format " << format << "): " << r.get_processor_resources().at(2)->error_text);
		if (!ok) continue;
		CHECK(sink->frames.size() == (20000 + 1151) / 1152, "frame count preserved");
		{
			// batching: frames already queued behind the first share its launch and its wait
			const size_t launches = vol->batch_stats.waits;
			CHECK(vol->batch_stats.rounds == sink->frames.size(), "every frame went through a batch");
			CHECK(launches >= 1 && launches < sink->frames.size(), "volume node batches queued frames: " << launches << " launches for " << sink->frames.size() << " frames");
		}
		size_t pos = 0;
		bool same = true;
		for (auto& f : sink->frames)
		{
			const Frame_data* d = f->data();
			same = same && d->format == format && d->sample_rate == 48000 && d->pts == (int64_t)(double(pos) / 48000 * 1000000);
			for (int i = 0; i < d->nb_samples && same; i++)
				for (int c = 0; c < 2 && same; c++)
				{
					const float x = src->samples[(pos + i) * 2 + c];
					if (format == AV_SAMPLE_FMT_FLT) same = reinterpret_cast<const float*>(d->data[0])[i * 2 + c] == x * 0.5f;
					else if (format == AV_SAMPLE_FMT_FLTP) same = reinterpret_cast<const float*>(d->data[c])[i] == x * 0.5f;
					else
					{
						const int16_t q = (int16_t)std::lrintf(x * 32767.0f);
						same = reinterpret_cast<const int16_t*>(d->data[0])[i * 2 + c] == (int16_t)((float)q * 0.5f);
					}
				}
			pos += d->nb_samples;
		}
		CHECK(same, "gain output bit-exact, metadata cloned (format " << format << ")");
	}
}

static void test_gpu_amix()
{
	const int S = 1152 * 9 + 300;
	Runner r;
	auto a = std::make_shared<Test_source>(), b = std::make_shared<Test_source>();
	a->samples = uniform(S * 2, 5);
	b->samples = uniform(S * 2, 6);
	b->format = AV_SAMPLE_FMT_FLTP;
	auto mix = std::make_shared<Audio_amix>();
	Json::Value v;
	v["input_num"] = 2;
	v["volumes0"] = 0.25; v["locks0"] = false;
	v["volumes1"] = 0.75; v["locks1"] = false;
	mix->deserialize(v);
	auto sink = std::make_shared<Test_sink>();
	r.add_node(1, a); r.add_node(2, b); r.add_node(3, mix); r.add_node(4, sink);
	r.add_link({1, "output", 3, "input_1"});
	r.add_link({2, "output", 3, "input_2"});
	r.add_link({3, "output", 4, "input"});
	const bool ok = r.run();
	CHECK(ok, "amix graph runs: " << r.get_processor_resources().at(3)->error_text);
	if (!ok) return;
	std::vector<float> aL(S), aR(S), bL(S), bR(S), oL(S), oR(S);
	for (int i = 0; i < S; i++) { aL[i] = a->samples[2 * i]; aR[i] = a->samples[2 * i + 1]; bL[i] = b->samples[2 * i]; bR[i] = b->samples[2 * i + 1]; }
	const float* inL[2] = {aL.data(), bL.data()};
	const float* inR[2] = {aR.data(), bR.data()};
	const float vol[2] = {0.25f, 0.75f};
	orc_amix_f32(inL, inR, vol, 2, oL.data(), oR.data(), S);
	size_t pos = 0;
	bool same = true;
	double t = 0;
	for (auto& f : sink->frames)
	{
		const Frame_data* d = f->data();
		if (pos >= (size_t)S) { for (int i = 0; i < d->nb_samples; i++) same = same && reinterpret_cast<const float*>(d->data[0])[i] == 0.0f; continue; }  // flush frames are silence
		same = same && d->format == AV_SAMPLE_FMT_FLTP && d->ch_layout.nb_channels == 2;
		t += d->nb_samples / 48000.0;
		same = same && d->pts == (int64_t)(t * 1000000);  // pts = END time (audio-amix.cpp:199-200)
		const int n = (int)std::min<size_t>(d->nb_samples, S - pos);
		same = same && std::memcmp(d->data[0], oL.data() + pos, n * sizeof(float)) == 0 && std::memcmp(d->data[1], oR.data() + pos, n * sizeof(float)) == 0;
		pos += d->nb_samples;
	}
	CHECK(pos >= (size_t)S, "all samples mixed (" << pos << ")");
	CHECK(same, "amix output bit-exact vs oracle, planar, pts = cumulative end time");
	// the sources fill their streams before the mixer's fiber runs: the waiting frames are mixed as batches behind one wait each
	CHECK(mix->batch_stats.rounds >= sink->frames.size() && mix->batch_stats.waits * 2 <= mix->batch_stats.rounds,
		  "amix batches the waiting frames: " << mix->batch_stats.rounds << " rounds behind " << mix->batch_stats.waits << " waits");
}

static void test_gpu_amix_converted_input()
{
	// input 1: 48 kHz stereo FLT (identity); input 2: 44.1 kHz MONO s16 -> converted on the GPU (N2)
	const int S = 1152 * 6;
	const int S2 = (int)(S * 44100.0 / 48000.0);
	Runner r;
	auto a = std::make_shared<Test_source>(), b = std::make_shared<Test_source>();
	a->samples = uniform(S * 2, 15);
	b->samples = uniform(S2, 16);
	b->ch = 1;
	b->format = AV_SAMPLE_FMT_S16;
	b->sample_rate = 44100;
	b->frame_size = 1024;
	auto mix = std::make_shared<Audio_amix>();
	Json::Value v;
	v["input_num"] = 2;
	v["volumes0"] = 0.5; v["locks0"] = false;
	v["volumes1"] = 0.5; v["locks1"] = false;
	mix->deserialize(v);
	auto sink = std::make_shared<Test_sink>();
	r.add_node(1, a); r.add_node(2, b); r.add_node(3, mix); r.add_node(4, sink);
	r.add_link({1, "output", 3, "input_1"});
	r.add_link({2, "output", 3, "input_2"});
	r.add_link({3, "output", 4, "input"});
	const bool ok = r.run();
	CHECK(ok, "amix with a 44.1 kHz mono s16 input runs: " << r.get_processor_resources().at(3)->error_text);
	if (!ok) return;
	// expected second input: K6 (s16 / 32768) -> m/sqrt(2) on both channels -> swr's default polyphase resampler 44.1k -> 48k
	std::vector<int16_t> q(S2);
	for (int i = 0; i < S2; i++) q[i] = (int16_t)std::lrintf(b->samples[i] * 32767.0f);
	std::vector<float> f(S2), st(S2);
	const void* pl[1] = {q.data()};
	orc_to_f32_interleaved(ORC_FMT_S16, pl, S2, 1, f.data());
	for (int i = 0; i < S2; i++) st[i] = f[i] * 0.70710678118654752440f;
	orc_swr_plan swr;
	orc_swr_plan_make(44100, 48000, &swr);
	std::vector<float> bank((size_t)swr.phase_count * swr.filter_alloc);
	orc_swr_build_filter(&swr, bank.data());
	struct { size_t out_len; } plan{orc_swr_out_len(&swr, S2)};
	std::vector<float> mono(plan.out_len), conv(plan.out_len * 2);
	orc_swr_resample_f32(&swr, bank.data(), st.data(), S2, 1, mono.data(), 1);
	for (size_t i = 0; i < plan.out_len; i++) conv[2 * i] = conv[2 * i + 1] = mono[i];
	// walk the mixer output: every sample = a*0.5 + b*0.5 with b from `conv` (zero once it is exhausted)
	size_t pos = 0;
	bool same = true;
	for (auto& fr : sink->frames)
	{
		const Frame_data* d = fr->data();
		for (int i = 0; i < d->nb_samples && same; i++, pos++)
			for (int c = 0; c < 2 && same; c++)
			{
				const float xa = pos < (size_t)S ? a->samples[2 * pos + c] : 0.0f;
				const float xb = pos < plan.out_len ? conv[2 * pos + c] : 0.0f;
				float acc = 0.0f;
				acc += xa * 0.5f;
				acc += xb * 0.5f;
				same = reinterpret_cast<const float*>(d->data[c])[i] == acc;
			}
	}
	CHECK(pos >= (size_t)S, "mixed at least the longer input: " << pos);
	CHECK(same, "amix output bit-exact vs oracle composition with a converted (44.1 kHz mono s16) input");
}

static void test_gpu_pitch_spectrum_fanout()
{
	const int S = 30000;
	const float semis = 3.0f;
	Runner r;
	auto src = std::make_shared<Test_source>();
	src->samples = uniform(S * 2, 7);
	auto pitch = std::make_shared<Pitch_modifier>();
	Json::Value v;
	v["pitch"] = (double)semis;
	pitch->deserialize(v);
	auto spec = std::make_shared<Audio_spectrum>();
	auto sink_audio = std::make_shared<Test_sink>(), sink_spec = std::make_shared<Test_sink>();
	r.add_node(1, src); r.add_node(2, pitch); r.add_node(3, spec); r.add_node(4, sink_audio); r.add_node(5, sink_spec);
	r.add_link({1, "output", 2, "input"});
	r.add_link({2, "output", 3, "input"});   // fan-out: the pitch node pushes each frame to two streams
	r.add_link({2, "output", 4, "input"});
	r.add_link({3, "output", 5, "input"});
	const bool ok = r.run();
	CHECK(ok, "pitch->spectrum graph runs: " << r.get_processor_resources().at(2)->error_text << r.get_processor_resources().at(3)->error_text);
	if (!ok) return;
	const float p = std::pow(2.0f, semis / 12.0f);  // what Pitch_modifier passes (audio-velocity.cpp:474)
	std::vector<float> ref(S * 2);
	CHECK(orc_stretch_f32(src->samples.data(), S, 2, 1.0, (double)p, ref.data()) == 0, "oracle stretch");
	const auto got = sink_audio->interleaved();
	CHECK(got.size() == ref.size(), "pitch output length " << got.size());
	CHECK(rel_rms(got, ref) <= 1e-4, "pitch output within 1e-4 RMS of the oracle: " << rel_rms(got, ref));
	const size_t F = orc_spectrum_frames(S);
	std::vector<float> sref(F * 2 * 513);
	orc_spectrum_f32(got.data(), S, 2, sref.data());
	CHECK(sink_spec->frames.size() == F, "spectrum frames " << sink_spec->frames.size() << " vs " << F);
	bool same = sink_spec->frames.size() == F;
	for (size_t f = 0; f < F && same; f++)
	{
		const Frame_data* d = sink_spec->frames[f]->data();
		same = d->nb_samples == 513 && d->format == AV_SAMPLE_FMT_FLTP;
		for (int c = 0; c < 2 && same; c++) same = std::memcmp(d->data[c], &sref[(f * 2 + c) * 513], 513 * sizeof(float)) == 0;
	}
	CHECK(same, "spectrum frames bit-exact vs oracle on the same input");
	for (auto& f : sink_audio->frames) CHECK(f->data()->nb_samples <= 3456, "chunks no larger than 3*1152/velocity (audio-velocity.cpp:417)");
	CHECK(pitch->batch_stats.rounds == (size_t)(S + 1151) / 1152 && pitch->batch_stats.waits * 2 <= pitch->batch_stats.rounds,
		  "pitch node puts the waiting frames as batches: " << pitch->batch_stats.rounds << " frames behind " << pitch->batch_stats.waits << " waits");
}

static void test_gpu_velocity_keep_pitch()
{
	const int S = 24000;
	Runner r;
	auto src = std::make_shared<Test_source>();
	src->samples = uniform(S * 2, 9);
	src->format = AV_SAMPLE_FMT_FLTP;
	auto vel = std::make_shared<Velocity_modifier>();
	Json::Value v;
	v["velocity"] = 1.5;
	v["keep_pitch"] = true;
	vel->deserialize(v);
	auto sink = std::make_shared<Test_sink>();
	r.add_node(1, src); r.add_node(2, vel); r.add_node(3, sink);
	r.add_link({1, "output", 2, "input"});
	r.add_link({2, "output", 3, "input"});
	CHECK(r.run(), "velocity graph runs: " << r.get_processor_resources().at(2)->error_text);
	const float velocity = 1.5f;
	std::vector<float> ref(16000 * 2);
	CHECK(orc_stretch_f32(src->samples.data(), S, 2, (double)velocity, (double)(1 / velocity), ref.data()) == 0, "oracle");
	const auto got = sink->interleaved();
	CHECK(got.size() == ref.size(), "length S/velocity: " << got.size());
	CHECK(rel_rms(got, ref) <= 1e-4, "velocity(keep pitch) within 1e-4 RMS: " << rel_rms(got, ref));
}

// "algorithm": "soundtouch" -> the WSOLA chain; frame-by-frame feeding gives the stream the oracle gives for the
// same put sequence, bit for bit (the chain is chunk-invariant: tests/test_wsola_cpu.py)
static void test_gpu_pitch_soundtouch_algorithm()
{
	const int S = 30000;
	const float semis = 3.0f;
	Runner r;
	auto src = std::make_shared<Test_source>();
	src->samples = uniform(S * 2, 21);
	auto pitch = std::make_shared<Pitch_modifier>();
	Json::Value v;
	v["pitch"] = (double)semis;
	v["algorithm"] = "soundtouch";
	pitch->deserialize(v);
	CHECK(pitch->serialize()["algorithm"].asString() == "soundtouch", "algorithm key round-trips");
	auto sink = std::make_shared<Test_sink>();
	r.add_node(1, src); r.add_node(2, pitch); r.add_node(3, sink);
	r.add_link({1, "output", 2, "input"});
	r.add_link({2, "output", 3, "input"});
	CHECK(r.run(), "soundtouch-algorithm graph runs: " << r.get_processor_resources().at(2)->error_text);
	const float p = std::pow(2.0f, semis / 12.0f);
	orc_st* st = nullptr;
	CHECK(orc_st_create(48000, 2, 1.0, (double)p, &st) == 0, "oracle create");
	for (int a = 0; a < S; a += src->frame_size)
		orc_st_put(st, src->samples.data() + 2 * a, (size_t)std::min(src->frame_size, S - a));
	orc_st_flush(st);
	std::vector<float> ref(orc_st_available(st) * 2);
	orc_st_receive(st, ref.data(), ref.size() / 2);
	orc_st_destroy(st);
	const auto got = sink->interleaved();
	CHECK(got.size() == ref.size() && got.size() == (size_t)S * 2, "soundtouch-algorithm output length " << got.size() << " vs " << ref.size());
	CHECK(got.size() == ref.size() && std::equal(got.begin(), got.end(), ref.begin()), "soundtouch-algorithm output is bit-exact vs the oracle chain");
}

// every running node owns its context — the spectrum node too (it ran on the process-wide fallback context, so two
// spectrum nodes of a graph shared one stream and waited for each other's work)
static void test_gpu_two_spectrum_nodes_have_own_contexts()
{
	const int S = 20000;
	Runner r;
	auto src = std::make_shared<Test_source>();
	src->samples = uniform(S * 2, 41);
	auto spec_a = std::make_shared<Audio_spectrum>(), spec_b = std::make_shared<Audio_spectrum>();
	auto sink_a = std::make_shared<Test_sink>(), sink_b = std::make_shared<Test_sink>();
	r.add_node(1, src); r.add_node(2, spec_a); r.add_node(3, spec_b); r.add_node(4, sink_a); r.add_node(5, sink_b);
	r.add_link({1, "output", 2, "input"});
	r.add_link({1, "output", 3, "input"});
	r.add_link({2, "output", 4, "input"});
	r.add_link({3, "output", 5, "input"});
	const size_t nodes_before = processor::gpu::flight_stats().nodes;
	const bool ok = r.run();
	CHECK(ok, "two spectrum nodes run: " << r.get_processor_resources().at(2)->error_text << r.get_processor_resources().at(3)->error_text);
	if (!ok) return;
	CHECK(spec_a->last_context != nullptr && spec_b->last_context != nullptr && spec_a->last_context != spec_b->last_context,
		  "the two spectrum nodes ran on different contexts");
	CHECK(processor::gpu::flight_stats().nodes == nodes_before + 2, "both are counted as GPU nodes");
	const size_t F = orc_spectrum_frames(S);
	std::vector<float> sref(F * 2 * 513);
	orc_spectrum_f32(src->samples.data(), S, 2, sref.data());
	for (const auto& sink : {sink_a, sink_b})
	{
		bool same = sink->frames.size() == F;
		for (size_t f = 0; f < F && same; f++)
			for (int c = 0; c < 2 && same; c++) same = std::memcmp(sink->frames[f]->data()->data[c], &sref[(f * 2 + c) * 513], 513 * sizeof(float)) == 0;
		CHECK(same, "each spectrum node's frames are bit-exact vs the oracle");
	}
}

// the pitch node's chunk sizes and time stamps (cadence): with frames arriving ONE AT A TIME the mirror emits the reference's own turn
// sequence (audio-velocity.cpp:399-436 restated in run_cadence against the oracle's SoundTouch-shaped handle) up to the flush frame;
// with frames arriving in batches the boundaries differ — the samples never do
static void test_gpu_pitch_cadence()
{
	const int S = 120000;
	const float semis = 3.0f;
	const float p = std::pow(2.0f, semis / 12.0f);
	const auto x = uniform((size_t)S * 2, 57);
	const Cadence ref = oracle_cadence(x, 1152, 1.0, (double)p, 1.0f, false, 1);
	for (int paced = 1; paced >= 0; paced--)
	{
		Runner r;
		auto src = std::make_shared<Test_source>();
		src->samples = x;
		src->one_at_a_time = paced != 0;
		auto pitch = std::make_shared<Pitch_modifier>();
		Json::Value v;
		v["pitch"] = (double)semis;
		v["algorithm"] = "soundtouch";
		pitch->deserialize(v);
		auto sink = std::make_shared<Test_sink>();
		sink->lazy_consumer = false;
		r.add_node(1, src); r.add_node(2, pitch); r.add_node(3, sink);
		r.add_link({1, "output", 2, "input"});
		r.add_link({2, "output", 3, "input"});
		const bool ok = r.run();
		CHECK(ok, "cadence graph runs: " << r.get_processor_resources().at(2)->error_text);
		if (!ok) return;
		size_t total = 0, same = 0;
		for (auto& f : sink->frames) total += (size_t)f->data()->nb_samples;
		CHECK(total == ref.total, "cadence: " << total << " samples delivered, the reference's turns deliver " << ref.total);
		while (same < ref.sizes.size() && same < sink->frames.size() && (size_t)sink->frames[same]->data()->nb_samples == ref.sizes[same] &&
			   sink->frames[same]->data()->pts == ref.pts[same])
			same++;
		if (paced)
			CHECK(same + 1 >= ref.sizes.size(), "frames arriving one at a time: the mirror's chunks and pts are the reference's up to the flush frame (" << same << " of "
				  << ref.sizes.size() << ", " << pitch->batch_stats.rounds << " frames behind " << pitch->batch_stats.waits << " waits)");
		else
		{
			const processor::cadence::Bounds b = processor::cadence::bounds(1.0f);
			bool inside = true;
			for (auto& f : sink->frames) inside = inside && (size_t)f->data()->nb_samples <= b.max_samples;
			CHECK(inside && sink->frames.size() <= ref.sizes.size() + 2, "frames arriving in batches: chunks of at most max, no more frames than the reference's ("
				  << sink->frames.size() << " vs " << ref.sizes.size() << ")");
		}
		// the float-typed microsecond clock: pts of frame k = float(sum of the samples in front of it / rate * 1e6)
		double t = 0.0;
		bool pts_ok = true;
		for (auto& f : sink->frames)
		{
			pts_ok = pts_ok && f->data()->pts == (int64_t)(float)(t * 1000000);
			t += double(f->data()->nb_samples) / 48000;
		}
		CHECK(pts_ok, "pts follow construct_audio_frame_float's float clock");
	}
}

static void test_gpu_bimix_v2()
{
	const int S = 5000;
	// left starts at t = 0, right 10 ms later: the first 480 output frames carry the left channel only
	Runner r;
	auto l = std::make_shared<Test_source>(), rr = std::make_shared<Test_source>();
	l->samples = uniform(S * 2, 11);
	rr->samples = uniform(S * 2, 12);
	rr->start_seconds = 0.010;
	auto mix = std::make_shared<Audio_bimix_v2>();
	auto sink = std::make_shared<Test_sink>();
	r.add_node(1, l); r.add_node(2, rr); r.add_node(3, mix); r.add_node(4, sink);
	r.add_link({1, "output", 3, "input_l"});
	r.add_link({2, "output", 3, "input_r"});
	r.add_link({3, "output", 4, "input"});
	const bool ok = r.run();
	CHECK(ok, "bimix_v2 graph runs: " << r.get_processor_resources().at(3)->error_text);
	if (!ok) return;
	std::vector<float> lm(S), rm(S);
	for (int i = 0; i < S; i++) { lm[i] = (float)((l->samples[2 * i] + l->samples[2 * i + 1]) * 0.5); rm[i] = (float)((rr->samples[2 * i] + rr->samples[2 * i + 1]) * 0.5); }
	const auto got = sink->interleaved();
	// walk the output: channel 0 must reproduce lm in order, channel 1 must reproduce rm in order after zeros
	size_t il = 0, ir = 0;
	bool same = true;
	size_t lead = 0;
	for (size_t i = 0; i < got.size() / 2; i++)
	{
		const float a = got[2 * i], b = got[2 * i + 1];
		if (il < (size_t)S && a == lm[il]) il++; else same = same && a == 0.0f;
		if (ir < (size_t)S && b == rm[ir]) ir++; else { same = same && b == 0.0f; if (ir == 0) lead++; }
	}
	CHECK(same, "every sample is either the next downmix sample of its side or silence");
	CHECK(il == (size_t)S, "left side complete: " << il);
	CHECK(ir + 2 >= (size_t)S, "right side complete up to the rounding slack the reference allows (audio-bimix.cpp:826): " << ir);
	CHECK(lead >= 478 && lead <= 482, "right channel starts ~480 frames late: " << lead);
	CHECK(mix->batch_stats.waits * 2 <= mix->batch_stats.rounds,
		  "bimix_v2 batches its intake and its output frames: " << mix->batch_stats.rounds << " frames behind " << mix->batch_stats.waits << " waits");
}

// velocity_modifier / pitch_modifier nodes without an "algorithm" key (projects saved by the reference) follow the default the
// integrator picked at registration; an explicit key wins; only a non-default choice is written back
static void test_velocity_cadence_rules()
{
	using namespace processor::cadence;
	// the rule itself (:416-423)
	const Bounds b1 = bounds(1.0f), b2 = bounds(2.0f), bh = bounds(0.5f), b3 = bounds(1.5f);
	CHECK(b1.min_samples == 1152 && b1.max_samples == 3456, "bounds at velocity 1");
	CHECK(b2.min_samples == 576 && b2.max_samples == 1728, "bounds at velocity 2");
	CHECK(bh.min_samples == 2304 && bh.max_samples == 6912, "bounds at velocity 0.5");
	CHECK(b3.min_samples == (uint32_t)(double(1.0f / 1.5f) * 1152) && b3.max_samples == (uint32_t)(double(1.0f / 1.5f) * 1152 * 3), "bounds truncate the double product");
	CHECK(reference_receive(1152, b1) == 0 && reference_receive(1153, b1) == 1153 && reference_receive(5000, b1) == 3456, "one turn's receive");
	CHECK((drain(1152, 1152, b1).empty()) && (drain(8000, 1152, b1) == std::vector<size_t>{3456, 3456}) && (drain(8000, 0, b1) == std::vector<size_t>{3456, 3456, 1088}),
		  "the mirror's turn = the rule until it no longer fires");
	// the reference's turn sequence against the SoundTouch-shaped oracle handle, and the mirror's: frames one at a time give the SAME
	// chunks and time stamps; batches of 16 give other boundaries, the same samples, sizes inside (min, max]
	const int S = 200000;
	const auto x = uniform((size_t)S * 2, 33);
	struct Setting { double rate, pitch; float velocity; const char* name; };
	const Setting settings[] = {{1.0, std::pow(2.0, 3.0 / 12.0), 1.0f, "pitch +3"}, {1.0, std::pow(2.0, -5.0 / 12.0), 1.0f, "pitch -5"},
								{2.0, 0.5, 2.0f, "velocity 2 keep pitch"}, {0.5, 2.0, 0.5f, "velocity 0.5 keep pitch"}, {1.5, 1.0, 1.5f, "velocity 1.5"}};
	for (const auto& st : settings)
	{
		const Bounds b = bounds(st.velocity);
		const Cadence ref = oracle_cadence(x, 1152, st.rate, st.pitch, st.velocity, false, 1);
		const Cadence one = oracle_cadence(x, 1152, st.rate, st.pitch, st.velocity, true, 1);
		const Cadence bat = oracle_cadence(x, 1152, st.rate, st.pitch, st.velocity, true, 16);
		CHECK(ref.total == one.total && ref.total == bat.total && ref.total > 0, st.name << ": all cadences deliver the same number of samples " << ref.total);
		// everything in front of the flush: identical chunk for chunk (the flush remainder is ONE frame in the reference, <= max pieces in the mirror)
		size_t same = 0;
		while (same < ref.sizes.size() && same < one.sizes.size() && ref.sizes[same] == one.sizes[same] && ref.pts[same] == one.pts[same]) same++;
		CHECK(same + 1 >= ref.sizes.size(), st.name << ": frames arriving one at a time -> the reference's chunks and pts up to the flush frame (" << same << " of "
			  << ref.sizes.size() << " equal)");
		size_t tail_ref = 0, tail_one = 0;
		for (size_t k = same; k < ref.sizes.size(); k++) tail_ref += ref.sizes[k];
		for (size_t k = same; k < one.sizes.size(); k++) tail_one += one.sizes[k];
		CHECK(tail_ref == tail_one, st.name << ": the flush remainder holds the same samples");
		bool inside = true;
		for (size_t k = 0; k + 1 < bat.sizes.size(); k++) inside = inside && bat.sizes[k] <= b.max_samples && bat.sizes[k] > 0;
		for (size_t k = 0; k < ref.sizes.size(); k++)
			if (k + 1 < ref.sizes.size()) inside = inside && ref.sizes[k] > b.min_samples && ref.sizes[k] <= b.max_samples;
		CHECK(inside, st.name << ": steady-state chunks lie in (min, max]");
		// (the flush remainder, ONE frame in the reference, is cut into pieces of at most max by the mirror: up to two more frames there)
		CHECK(bat.sizes.size() <= ref.sizes.size() + 2, st.name << ": batching makes larger chunks, not more frames (" << bat.sizes.size() << " vs " << ref.sizes.size() << ")");
	}
}

static void test_default_stretch_algorithm()
{
	using namespace processor;
	Json::Value none, st, voc;
	st["algorithm"] = "soundtouch";
	voc["algorithm"] = "vocoder";
	none["pitch"] = 3.0; st["pitch"] = 3.0; voc["pitch"] = 3.0;
	CHECK(default_stretch_algorithm() == Stretch_algorithm::Vocoder, "library default is the vocoder");
	CHECK(algorithm_from_json(none) == Stretch_algorithm::Vocoder && algorithm_from_json(st) == Stretch_algorithm::Soundtouch, "keys under the vocoder default");
	set_default_stretch_algorithm(Stretch_algorithm::Soundtouch);
	CHECK(algorithm_from_json(none) == Stretch_algorithm::Soundtouch && algorithm_from_json(voc) == Stretch_algorithm::Vocoder, "keys under the soundtouch default");
	{
		Pitch_modifier p;                       // a fresh node takes the default ...
		p.deserialize(none);
		CHECK(!p.serialize().isMember("algorithm"), "the default choice is not written back (the reference's JSON stays unchanged)");
		p.deserialize(voc);                     // ... an explicit choice that differs from it is kept
		CHECK(p.serialize().isMember("algorithm") && p.serialize()["algorithm"].asString() == "vocoder", "a non-default choice is serialised");
	}
	set_default_stretch_algorithm(Stretch_algorithm::Vocoder);
}

// the pts-alignment rule of Audio_bimix_v2 as a pure function (host/processor/bimix-align.hpp), case by case against
// /root/reference/src/processor/audio-bimix.cpp:777-872 worked by hand
static void test_bimix_align_step()
{
	using processor::bimix::Span;
	using processor::bimix::align_step;
	const int R = 48000;
	{   // disjoint: the left block ends before the right one begins -> left alone, used up
		const auto s = align_step(Span{0.0, 480}, Span{0.010, 1000}, R);
		CHECK(s.first == 0 && s.solo == 480 && s.both == 0 && s.used_up[0] && !s.used_up[1] && s.played[1] == 0, "disjoint spans");
	}
	{   // right begins 10 ms into the left block and outlasts it: 480 solo + 520 together; left used up, right loses 520
		const auto s = align_step(Span{0.0, 1000}, Span{0.010, 1152}, R);
		CHECK(s.first == 0 && s.solo == 480 && s.both == 520 && s.used_up[0] && !s.used_up[1] && s.played[1] == 520, "overlap, first ends first");
	}
	{   // the later block ends first: it is used up, the earlier one loses solo + both
		const auto s = align_step(Span{0.0, 4000}, Span{0.010, 1000}, R);
		CHECK(s.first == 0 && s.solo == 480 && s.both == 1000 && !s.used_up[0] && s.used_up[1] && s.played[0] == 1480, "overlap, later ends first");
	}
	{   // equal start: the RIGHT side counts as first (the reference tests left < right), nothing solo
		const auto s = align_step(Span{1.5, 1152}, Span{1.5, 1152}, R);
		CHECK(s.first == 1 && s.solo == 0 && s.both == 1152 && s.used_up[1] && !s.used_up[0] && s.played[0] == 1152, "simultaneous spans");
	}
	{   // right first
		const auto s = align_step(Span{0.5, 1152}, Span{0.49, 1152}, R);
		CHECK(s.first == 1 && s.solo == 480 && s.both == 672 && s.used_up[1] && s.played[0] == 672, "right side first");
	}
	{   // rounding can ask for one sample more than a block holds: limited by both blocks
		const auto s = align_step(Span{0.0, 100}, Span{0.0010312, 51}, R);      // 49.4976 samples later -> solo 49, both min(51, 51, 51)
		CHECK(s.first == 0 && s.solo == 49 && s.both == 51 && s.solo + s.both <= 100, "limits");
	}
}

// Audio_bimix (v1) through the runner: left input -> left channel, right input -> right channel, bias (audio-bimix.cpp:83-331,
// inner loop :310-317), against the oracle's K4
static void test_gpu_bimix_v1()
{
	const int S = 1152 * 5;
	Runner r;
	auto l = std::make_shared<Test_source>(), rr = std::make_shared<Test_source>();
	l->samples = uniform(S * 2, 21);
	rr->samples = uniform(S * 2, 22);
	auto mix = std::make_shared<Audio_bimix>();
	Json::Value v;
	v["bias"] = 0.25;
	mix->deserialize(v);
	auto sink = std::make_shared<Test_sink>();
	r.add_node(1, l); r.add_node(2, rr); r.add_node(3, mix); r.add_node(4, sink);
	r.add_link({1, "output", 3, "input_l"});
	r.add_link({2, "output", 3, "input_r"});
	r.add_link({3, "output", 4, "input"});
	const bool ok = r.run();
	CHECK(ok, "bimix graph runs: " << r.get_processor_resources().at(3)->error_text);
	if (!ok) return;
	std::vector<float> ll(S), lr(S), rl(S), rrr(S), oL(S), oR(S);
	for (int i = 0; i < S; i++) { ll[i] = l->samples[2 * i]; lr[i] = l->samples[2 * i + 1]; rl[i] = rr->samples[2 * i]; rrr[i] = rr->samples[2 * i + 1]; }
	orc_bimix_f32(ll.data(), lr.data(), rl.data(), rrr.data(), 0.25f, oL.data(), oR.data(), S);
	size_t pos = 0;
	bool same = true;
	for (auto& fr : sink->frames)
	{
		const Frame_data* d = fr->data();
		CHECK(d->format == AV_SAMPLE_FMT_FLTP && d->ch_layout.nb_channels == 2, "bimix delivers planar stereo float");
		for (int i = 0; i < d->nb_samples && pos < (size_t)S; i++, pos++)
			same = same && reinterpret_cast<const float*>(d->data[0])[i] == oL[pos] && reinterpret_cast<const float*>(d->data[1])[i] == oR[pos];
	}
	CHECK(pos == (size_t)S, "bimix delivered every sample: " << pos);
	CHECK(same, "bimix output bit-exact vs the oracle (K4)");
	CHECK(mix->batch_stats.waits * 2 <= mix->batch_stats.rounds,
		  "bimix batches the waiting frames: " << mix->batch_stats.rounds << " rounds behind " << mix->batch_stats.waits << " waits");
}

// ------------------------------------------------------------------------------------------------ plugin-boundary benchmark
// What the editor sees (include/infra/processor.hpp:108-113: process_payload; the loop it runs, src/processor/audio-vol.cpp:137-150):
// 1152-sample frames of 48 kHz stereo f32 pushed through  audio_volume_adjust -> audio_amix(2) -> pitch_modifier  by the fiber
// runner on ONE thread, every hop a host frame (upload, kernels, download).  `branches` independent copies of that graph share the
// runner, as independent tracks of a project would.  Prints one line:  HOST_PATH {json}
#include <chrono>
#include <thread>
static void bench_host_path(int branches, double seconds, bool print)
{
	const int S = (int)(seconds * 48000);
	const float semis = 3.0f, vol = 0.7071f;
	Runner r;
	std::vector<std::shared_ptr<Test_source>> srcs;
	std::vector<std::shared_ptr<Test_sink>> sinks;
	int id = 1;
	for (int b = 0; b < branches; b++)
	{
		auto a = std::make_shared<Test_source>(), c = std::make_shared<Test_source>();
		a->samples = uniform((size_t)S * 2, 100 + 2 * b);
		c->samples = uniform((size_t)S * 2, 101 + 2 * b);
		auto gain = std::make_shared<Audio_vol>();
		gain->set_volume(vol);
		auto mix = std::make_shared<Audio_amix>();
		Json::Value v;
		v["input_num"] = 2;
		v["volumes0"] = 0.5; v["locks0"] = false;
		v["volumes1"] = 0.5; v["locks1"] = false;
		mix->deserialize(v);
		auto pitch = std::make_shared<Pitch_modifier>();
		Json::Value pv;
		pv["pitch"] = (double)semis;
		pitch->deserialize(pv);
		auto sink = std::make_shared<Test_sink>();
		sink->lazy_consumer = false;
		sink->keep = b == 0;
		const int n_a = id++, n_c = id++, n_g = id++, n_m = id++, n_p = id++, n_s = id++;
		r.add_node(n_a, a); r.add_node(n_c, c); r.add_node(n_g, gain); r.add_node(n_m, mix); r.add_node(n_p, pitch); r.add_node(n_s, sink);
		r.add_link({n_a, "output", n_g, "input"});
		r.add_link({n_g, "output", n_m, "input_1"});
		r.add_link({n_c, "output", n_m, "input_2"});
		r.add_link({n_m, "output", n_p, "input"});
		r.add_link({n_p, "output", n_s, "input"});
		srcs.push_back(a); srcs.push_back(c);
		sinks.push_back(sink);
	}
	const int devices_before = gpu::flight_stats().devices_used;
	gpu::flight_stats() = gpu::Flight_stats{};
	gpu::flight_stats().devices_used = devices_before;
	// watchdog: a graph that has not finished after 60 s is stopped and the nodes still running are named (never hang the box)
	std::atomic<bool> finished{false};
	std::string stuck;
	std::thread watchdog([&] {
		for (int i = 0; i < 600 && !finished; i++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
		if (finished) return;
		for (auto& [nid, res] : r.get_processor_resources())
			if (res->state == Runner::State::Running) stuck += " " + std::to_string(nid) + ":" + res->processor->get_processor_info_non_static().identifier;
		r.request_stop();
	});
	const auto t0 = std::chrono::steady_clock::now();
	const bool ok = r.run();
	const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	finished = true;
	watchdog.join();
	CHECK(stuck.empty(), "host-path benchmark graph finished by itself; still running after 60 s:" << stuck);
	CHECK(ok, "host-path benchmark graph runs");
	if (!stuck.empty()) { std::cout << std::flush; return; }
	if (!ok)
	{
		for (auto& [nid, res] : r.get_processor_resources())
			if (!res->error_text.empty()) std::cout << "  node " << nid << ": " << res->error_text << "\n";
		return;
	}
	const gpu::Flight_stats fs = gpu::flight_stats();
	size_t out_samples = 0;
	for (auto& s : sinks) out_samples += s->n_samples;
	CHECK(out_samples >= (size_t)branches * (size_t)(S - 2), "every branch delivered its audio (" << out_samples << " sample-frames)");
	// the CPU oracle on the same frames, one thread (the reference runs the graph on one thread: src/infra/runner.cpp:65-69)
	double cpu = 0.0;
	double err = 0.0;
	{
		const auto c0 = std::chrono::steady_clock::now();
		std::vector<float> g((size_t)S * 2), gL(S), gR(S), cL(S), cR(S), oL(S), oR(S), m((size_t)S * 2), ref((size_t)S * 2);
		for (int b = 0; b < branches; b++)
		{
			const float* sp[1] = {srcs[2 * b]->samples.data()};
			float* dp[1] = {g.data()};
			orc_change_volume_f32(dp, sp, 1, S * 2, vol);
			const std::vector<float>& c = srcs[2 * b + 1]->samples;
			for (int i = 0; i < S; i++) { gL[i] = g[2 * i]; gR[i] = g[2 * i + 1]; cL[i] = c[2 * i]; cR[i] = c[2 * i + 1]; }
			const float* inL[2] = {gL.data(), cL.data()};
			const float* inR[2] = {gR.data(), cR.data()};
			const float w[2] = {0.5f, 0.5f};
			orc_amix_f32(inL, inR, w, 2, oL.data(), oR.data(), S);
			for (int i = 0; i < S; i++) { m[2 * i] = oL[i]; m[2 * i + 1] = oR[i]; }
			// the mixer ends its stream with flush frames of silence (audio-amix.cpp:281-292,320): the pitch node sees Sp >= S frames
			const size_t Sp = b == 0 ? std::max<size_t>(sinks[0]->n_samples, (size_t)S) : (size_t)S;
			m.resize(Sp * 2, 0.0f);
			ref.resize(Sp * 2);
			orc_stretch_f32(m.data(), Sp, 2, 1.0, (double)std::pow(2.0f, semis / 12.0f), ref.data());
			if (b == 0)
			{
				auto got = sinks[0]->interleaved();
				const size_t got_len = got.size();
				CHECK(got.size() == ref.size(), "pitch node delivers the length its input has: " << got.size() / 2 << " vs " << ref.size() / 2);
				got.resize(ref.size());
				err = rel_rms(got, ref);
				if (err > 1e-6)
				{
					size_t first = ref.size(), last = 0, worst = 0;
					for (size_t i = 0; i < ref.size(); i++)
						if (std::fabs(got[i] - ref[i]) > 1e-4f)
						{
							if (first == ref.size()) first = i;
							last = i;
							if (std::fabs(got[i] - ref[i]) > std::fabs(got[worst] - ref[worst])) worst = i;
						}
					std::cout << "  branch 0: " << got_len / 2 << " frames delivered for " << ref.size() / 2 << "; |diff| > 1e-4 from sample-frame " << first / 2
							  << " to " << last / 2 << ", worst " << std::fabs(got[worst] - ref[worst]) << " at " << worst / 2 << "\n";
				}
			}
		}
		cpu = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
	}
	CHECK(err <= 1e-4, "branch 0 of the benchmark graph within 1e-4 RMS of the oracle: " << err);
	if (!print) return;
	const double in_frames = (double)branches * 2.0 * ((S + 1151) / 1152);      // 1152-sample frames the sources pushed
	char line[1024];
	std::snprintf(line, sizeof line,
				  "HOST_PATH {\"graph\": \"audio_volume_adjust -> audio_amix(2) -> pitch_modifier(+3 st), 48 kHz stereo f32, 1152-sample frames\", "
				  "\"branches\": %d, \"seconds_per_branch\": %.1f, \"wall_s\": %.4f, \"source_frames_per_s\": %.1f, \"sample_frames_per_s\": %.1f, "
				  "\"real_time_factor\": %.1f, \"setup_s\": %.4f, \"contexts_created\": %zu, \"sample_frames_per_s_without_setup\": %.1f, "
				  "\"gpu_nodes\": %zu, \"devices_used\": %d, \"waits\": %zu, \"waits_per_source_frame\": %.4f, "
				  "\"polls_per_wait\": %.2f, \"max_nodes_in_flight\": %d, \"fiber_switches\": %zu, \"cpu_oracle_s\": %.4f, "
				  "\"cpu_oracle_sample_frames_per_s\": %.1f, \"rel_rms_branch0\": %.3g}",
				  branches, seconds, wall, in_frames / wall, (double)branches * S / wall, (double)branches * seconds / wall, fs.setup_seconds,
				  fs.contexts_created, (double)branches * S / std::max(wall - fs.setup_seconds, 1e-9), fs.nodes, fs.devices_used,
				  fs.waits, fs.waits / in_frames, fs.waits ? (double)fs.polls / fs.waits : 0.0, fs.max_in_flight, r.context_switches(), cpu,
				  (double)branches * S / cpu, err);
	std::cout << line << "\n";
}

// ---- where the host thread's ~10 us per source frame go: the per-frame operations timed by themselves, and the
// benchmark graph grown node by node (each variant on `seconds` of 48 kHz stereo in 1152-sample frames, one runner thread)
static double run_chain(int n_vol, bool mix, bool pitch, double seconds, size_t* switches)
{
	const int S = (int)(seconds * 48000);
	Runner r;
	int id = 1;
	auto a = std::make_shared<Test_source>();
	a->samples = uniform((size_t)S * 2, 500);
	int prev = id++;
	r.add_node(prev, a);
	for (int k = 0; k < n_vol; k++)
	{
		auto g = std::make_shared<Audio_vol>();
		g->set_volume(0.9f);
		const int n = id++;
		r.add_node(n, g);
		r.add_link({prev, "output", n, "input"});
		prev = n;
	}
	if (mix)
	{
		auto c = std::make_shared<Test_source>();
		c->samples = uniform((size_t)S * 2, 501);
		auto m = std::make_shared<Audio_amix>();
		Json::Value v;
		v["input_num"] = 2;
		v["volumes0"] = 0.5; v["locks0"] = false;
		v["volumes1"] = 0.5; v["locks1"] = false;
		m->deserialize(v);
		const int nc = id++, nm = id++;
		r.add_node(nc, c); r.add_node(nm, m);
		r.add_link({prev, "output", nm, "input_1"});
		r.add_link({nc, "output", nm, "input_2"});
		prev = nm;
	}
	if (pitch)
	{
		auto pm = std::make_shared<Pitch_modifier>();
		Json::Value pv;
		pv["pitch"] = 3.0;
		pm->deserialize(pv);
		const int n = id++;
		r.add_node(n, pm);
		r.add_link({prev, "output", n, "input"});
		prev = n;
	}
	auto sink = std::make_shared<Test_sink>();
	sink->lazy_consumer = false;
	sink->keep = false;
	const int ns = id++;
	r.add_node(ns, sink);
	r.add_link({prev, "output", ns, "input"});
	const auto t0 = std::chrono::steady_clock::now();
	const bool ok = r.run();
	const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	CHECK(ok, "host-cost chain runs");
	if (switches) *switches = r.context_switches();
	return wall;
}

static void host_costs(double seconds)
{
	using clk = std::chrono::steady_clock;
	auto ns_each = [](clk::time_point t0, size_t n) { return std::chrono::duration<double, std::nano>(clk::now() - t0).count() / n; };
	const size_t N = 200000;
	// (a) one output frame: shared Audio_frame + 32-byte aligned buffer for 1152 stereo f32 samples, and its release
	auto t0 = clk::now();
	for (size_t i = 0; i < N; i++)
	{
		auto f = std::make_shared<Audio_frame>();
		Frame_data* d = f->data();
		d->format = AV_SAMPLE_FMT_FLT; d->nb_samples = 1152; d->ch_layout.nb_channels = 2;
		frame_get_buffer(d, 32);
		reinterpret_cast<volatile float*>(d->data[0])[0] = 1.0f;
	}
	const double alloc_ns = ns_each(t0, N);
	// (b) one 9216-byte copy (frame <-> page-locked staging), source and destination cycling through 16 MiB
	std::vector<uint8_t> src(16 << 20), dst(16 << 20);
	t0 = clk::now();
	for (size_t i = 0; i < N; i++) std::memcpy(dst.data() + (i * 9216) % ((16 << 20) - 9216), src.data() + (i * 9216 * 7) % ((16 << 20) - 9216), 9216);
	const double copy_ns = ns_each(t0, N);
	// (c) one frame through a stream: try_push + try_pop
	Audio_stream st;
	auto fr = std::make_shared<Audio_frame>();
	t0 = clk::now();
	for (size_t i = 0; i < N; i++) { st.try_push(fr); (void)st.try_pop(); }
	const double stream_ns = ns_each(t0, N);
	// (d) the graph grown node by node; host us per source frame = wall / frames (the GPU work of these nodes is far shorter than the wall time)
	const double frames = std::ceil(seconds * 48000 / 1152.0);
	size_t sw[5] = {0, 0, 0, 0, 0};
	run_chain(1, true, true, 2.0, nullptr);                      // untimed: contexts, first launches
	const double w0 = run_chain(0, false, false, seconds, &sw[0]);
	const double w1 = run_chain(1, false, false, seconds, &sw[1]);
	const double w2 = run_chain(2, false, false, seconds, &sw[2]);
	const double w3 = run_chain(1, true, false, seconds, &sw[3]);
	const double w4 = run_chain(1, true, true, seconds, &sw[4]);
	const double yield_ns = sw[0] ? w0 * 1e9 / sw[0] : 0.0;      // source -> sink only: almost nothing but stream operations and fiber switches
	char line[1024];
	std::snprintf(line, sizeof line,
				  "HOST_COSTS {\"frame_alloc_ns\": %.0f, \"copy_9216B_ns\": %.0f, \"stream_push_pop_ns\": %.0f, \"source_to_sink_ns_per_fiber_switch\": %.0f, "
				  "\"us_per_source_frame\": {\"source->sink\": %.2f, \"+volume\": %.2f, \"+volume+volume\": %.2f, \"+volume+amix(2)\": %.2f, "
				  "\"+volume+amix(2)+pitch\": %.2f}, \"fiber_switches\": [%zu, %zu, %zu, %zu, %zu], \"frames_per_chain\": %.0f}",
				  alloc_ns, copy_ns, stream_ns, yield_ns, w0 / frames * 1e6, w1 / frames * 1e6, w2 / frames * 1e6, w3 / frames * 1e6, w4 / frames * 1e6, sw[0], sw[1],
				  sw[2], sw[3], sw[4], frames);
	std::cout << line << "\n";
}

int main(int argc, char** argv)
{
	const std::string mode = argc > 1 ? argv[1] : "cpu";
	if (mode == "hostcost")
	{
		host_costs(argc > 2 ? std::atof(argv[2]) : 120.0);
		std::cout << (failures ? "SELFTEST FAILED " : "SELFTEST OK ") << mode << " failures=" << failures << "\n";
		return failures ? 1 : 0;
	}
	if (mode == "registry")
	{
		print_registry();
		return 0;
	}
	if (mode == "bench")
	{
		const double seconds = argc > 2 ? std::atof(argv[2]) : 20.0;
		bench_host_path(1, 2.0, false);     // untimed: module load, first-launch costs
		bench_host_path(1, seconds, true);
		bench_host_path(16, seconds, true);
		std::cout << (failures ? "SELFTEST FAILED " : "SELFTEST OK ") << mode << " failures=" << failures << "\n";
		return failures ? 1 : 0;
	}
	test_streams_and_scheduler();
	test_registry_and_json();
	test_headless_draw_hooks();
	test_bimix_align_step();
	test_default_stretch_algorithm();
	test_velocity_cadence_rules();
	if (mode == "gpu")
	{
		test_error_capture();
		test_gpu_volume();
		test_gpu_amix();
		test_gpu_amix_converted_input();
		test_gpu_pitch_spectrum_fanout();
		test_gpu_velocity_keep_pitch();
		test_gpu_pitch_soundtouch_algorithm();
		test_gpu_pitch_cadence();
		test_gpu_two_spectrum_nodes_have_own_contexts();
		test_gpu_bimix_v1();
		test_gpu_bimix_v2();
	}
	std::cout << (failures ? "SELFTEST FAILED " : "SELFTEST OK ") << mode << " failures=" << failures << "\n";
	return failures ? 1 : 0;
}
