// processor/audio-velocity.hpp — GPU drop-ins for processor::Velocity_modifier and processor::Pitch_modifier
// (/root/reference/include/processor/audio-velocity.hpp, src/processor/audio-velocity.cpp:265-505), plus the
// FFT spectrum node the reference lists as a feature (README.md:28) but never implemented.
#pragma once
#include "audio-stream.hpp"

namespace processor
{
	// Which GPU implementation stands in for SoundTouch: the phase vocoder BASELINE.json's north_star asks for, or the
	// WSOLA + anti-alias FIR + cubic transposer chain restated from SoundTouch 2.3.2 (nae_wsola_*).
	// A node's JSON may name it ("algorithm": "vocoder" | "soundtouch").  The reference's JSON has no such key
	// (audio-velocity.cpp:479-505), so projects saved by it — and freshly created nodes — get the DEFAULT, which the
	// integrator chooses once at registration: infra::register_all_processors(Stretch_algorithm::Soundtouch) keeps the
	// reference's audible behaviour for saved projects; the plain call (and this library's own default) is the vocoder.
	enum class Stretch_algorithm { Vocoder, Soundtouch };
	const char* algorithm_name(Stretch_algorithm a);
	Stretch_algorithm default_stretch_algorithm();
	void set_default_stretch_algorithm(Stretch_algorithm a);
	Stretch_algorithm algorithm_from_json(const Json::Value& value);

	class Velocity_modifier : public infra::Processor
	{
		float velocity = 1;
		bool keep_pitch = false;
		Stretch_algorithm algorithm = default_stretch_algorithm();

	  public:

		Batch_stats batch_stats;  // of the last process_payload: frames put / waits for their uploads

		static infra::Processor::Info get_processor_info();
		Processor::Info get_processor_info_non_static() const override { return get_processor_info(); }
		void draw_title() override;                         // bodies: draw-headless.cpp (the integrator keeps the reference's ImGui bodies instead)
		bool draw_content(bool readonly) override;
		std::vector<infra::Processor::Pin_attribute> get_pin_attributes() const override;
		void process_payload(
			const std::map<std::string, std::shared_ptr<infra::Processor::Product>>& input,
			const std::map<std::string, std::set<std::shared_ptr<infra::Processor::Product>>>& output,
			const std::atomic<bool>& stop_token,
			std::any& user_data
		) override;
		Json::Value serialize() const override;            // velocity, keep_pitch (audio-velocity.cpp:479-485)
		void deserialize(const Json::Value& value) override;  // :487-493
	};

	class Pitch_modifier : public infra::Processor
	{
		float pitch = 0;  // semitones
		Stretch_algorithm algorithm = default_stretch_algorithm();

	  public:

		Batch_stats batch_stats;  // of the last process_payload: frames put / waits for their uploads

		static infra::Processor::Info get_processor_info();
		Processor::Info get_processor_info_non_static() const override { return get_processor_info(); }
		void draw_title() override;                         // bodies: draw-headless.cpp (the integrator keeps the reference's ImGui bodies instead)
		bool draw_content(bool readonly) override;
		std::vector<infra::Processor::Pin_attribute> get_pin_attributes() const override;
		void process_payload(
			const std::map<std::string, std::shared_ptr<infra::Processor::Product>>& input,
			const std::map<std::string, std::set<std::shared_ptr<infra::Processor::Product>>>& output,
			const std::atomic<bool>& stop_token,
			std::any& user_data
		) override;
		Json::Value serialize() const override;            // pitch (:495-500)
		void deserialize(const Json::Value& value) override;  // :502-505
	};

	// New node (registered as "audio_spectrum"): per channel, Hann-windowed 1024-point r2c magnitude every 256
	// sample-frames.  Output stays an Audio_stream so the editor's pin type check passes: one FLTP frame per hop with
	// nb_samples = 513 (bins), plane c = |X_c[k]|, pts = start time of the analysed window.
	class Audio_spectrum : public infra::Processor
	{
	  public:

		const void* last_context = nullptr;  // the nae_ctx its last process_payload ran on (every running node owns one: gpu-context.hpp); tests only

		static infra::Processor::Info get_processor_info();
		Processor::Info get_processor_info_non_static() const override { return get_processor_info(); }
		void draw_title() override;                         // bodies: draw-headless.cpp (the integrator keeps the reference's ImGui bodies instead)
		bool draw_content(bool readonly) override;
		std::vector<infra::Processor::Pin_attribute> get_pin_attributes() const override;
		void process_payload(
			const std::map<std::string, std::shared_ptr<infra::Processor::Product>>& input,
			const std::map<std::string, std::set<std::shared_ptr<infra::Processor::Product>>>& output,
			const std::atomic<bool>& stop_token,
			std::any& user_data
		) override;
		Json::Value serialize() const override { return {}; }
		void deserialize(const Json::Value&) override {}
	};
}
