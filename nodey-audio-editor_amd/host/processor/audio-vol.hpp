// processor/audio-vol.hpp — GPU drop-in for processor::Audio_vol
// (/root/reference/include/processor/audio-vol.hpp:33-61, src/processor/audio-vol.cpp).
#pragma once
#include "audio-stream.hpp"

namespace processor
{
	class Audio_vol : public infra::Processor
	{
		float volume = 1.0;

	  public:

		Batch_stats batch_stats;  // of the last process_payload: frames scaled / waits (one launch and one wait per batch)

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
		// the reference does NOT serialise volume (audio-vol.hpp:57-58)
		Json::Value serialize() const override { return {}; }
		void deserialize(const Json::Value&) override {}
		// the GUI slider (audio-vol.cpp:266-274) clamps to [0, max_volume = 10]
		void set_volume(float v) { volume = v < 0 ? 0 : (v > 10.0f ? 10.0f : v); }
		float get_volume() const { return volume; }
	};
}
