// processor/draw-headless.cpp — the two GUI hooks of every GPU processor, without a GUI.
//
// In the reference draw_title() / draw_content(bool readonly) are pure virtual (/root/reference/include/infra/processor.hpp:99-104), run on the
// GUI thread once per displayed frame (src/frontend/app.cpp:264-265), and are the ONLY code that changes a node's parameters while the editor
// runs: `volume` is not even serialised (include/processor/audio-vol.hpp:57-58; slider and clamp: src/processor/audio-vol.cpp:257-281), the
// mixer's weights are renormalised there so that the unlocked ones sum to 1 (audio-amix.cpp:331-393), `bias` (audio-bimix.cpp:338-356),
// `velocity` / `keep_pitch` (audio-velocity.cpp:107-133) and `pitch` (:135-149) have no other writer besides deserialize().
//
// An integrator therefore does NOT link this file: the reference's own bodies of these functions (ImGui calls around the same member names —
// volume, input_num / volumes / locks, bias, velocity / keep_pitch, pitch) compile unchanged as members of the GPU classes and take its place.
// What stands here is what those bodies do to the parameters when no widget is touched — the clamps and the mixer's renormalisation — so that a
// headless host (tests/host/selftest, an export tool) that calls draw_content() once after editing parameters gets the values the editor
// would run with.  Every body returns what the reference returns when nothing changed: false (true = "pins changed", audio-amix.cpp:344-348).
#include <algorithm>

#include "audio-mix.hpp"
#include "audio-velocity.hpp"
#include "audio-vol.hpp"

namespace processor
{
	void Audio_vol::draw_title() {}
	bool Audio_vol::draw_content(bool)
	{
		set_volume(volume);  // [0, max_volume = 10]
		return false;
	}

	void Audio_amix::draw_title() {}
	bool Audio_amix::draw_content(bool)
	{
		input_num = std::clamp(input_num, 1, 16);
		volumes.resize(input_num, 1.0f);
		locks.resize(input_num, false);
		float unlocked = 0.0f;
		for (int i = 0; i < input_num; i++) unlocked += locks[i] ? 0.0f : volumes[i];
		unlocked = std::max(unlocked, 0.001f);
		for (int i = 0; i < input_num; i++)
			if (!locks[i]) volumes[i] /= unlocked;
		return false;
	}

	void Audio_bimix::draw_title() {}
	bool Audio_bimix::draw_content(bool)
	{
		bias = std::clamp(bias, -1.0f, 1.0f);
		return false;
	}

	void Audio_bimix_v2::draw_title() {}
	bool Audio_bimix_v2::draw_content(bool) { return false; }

	void Velocity_modifier::draw_title() {}
	bool Velocity_modifier::draw_content(bool)
	{
		velocity = std::clamp(velocity, 0.5f, 3.0f);  // ImGuiSliderFlags_AlwaysClamp
		return false;
	}

	void Pitch_modifier::draw_title() {}
	bool Pitch_modifier::draw_content(bool) { return false; }

	void Audio_spectrum::draw_title() {}
	bool Audio_spectrum::draw_content(bool) { return false; }
}
