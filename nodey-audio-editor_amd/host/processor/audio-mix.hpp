// processor/audio-mix.hpp — GPU drop-ins for processor::Audio_amix (/root/reference/include/processor/audio-amix.hpp,
// src/processor/audio-amix.cpp), Audio_bimix and Audio_bimix_v2 (include/processor/audio-bimix.hpp,
// src/processor/audio-bimix.cpp).  Same identifiers, pins and JSON keys, so saved projects keep loading
// (src/infra/graph.cpp:399-408).
//
// Input envelope: the reference first converts every input to 48 kHz stereo FLTP with libswresample.  That role is
// nae_swr here: 48 kHz stereo FLT/FLTP is a pure deinterleave (bit-exact, K2); other rates, mono and integer formats
// are converted on the GPU with the builder's own filter (SURVEY.md §8f N2 — unpinned versus FFmpeg).
#pragma once
#include "audio-stream.hpp"

namespace processor
{
	class Audio_amix : public infra::Processor
	{
		int input_num = 2;
		std::vector<float> volumes;
		std::vector<bool> locks;

	  public:

		Batch_stats batch_stats;  // of the last process_payload

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
		Json::Value serialize() const override;            // input_num, volumes{i}, locks{i}  (audio-amix.cpp:395-405)
		void deserialize(const Json::Value& value) override;  // :407-423
	};

	class Audio_bimix : public infra::Processor
	{
		float bias = 0;

	  public:

		Batch_stats batch_stats;  // of the last process_payload

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
		Json::Value serialize() const override;            // bias (audio-bimix.cpp:358-363)
		void deserialize(const Json::Value& value) override;  // :365-383
	};

	class Audio_bimix_v2 : public infra::Processor
	{
	  public:

		Batch_stats batch_stats;  // of the last process_payload: input frames taken + output frames built / waits

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
		Json::Value serialize() const override { return {}; }   // audio-bimix.cpp:444-449
		void deserialize(const Json::Value&) override {}
	};
}
