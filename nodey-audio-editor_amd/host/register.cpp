// register.cpp — fills Processor::processor_map with the GPU processors, under the identifiers of the CPU classes
// they replace (reference list: src/register.cpp:16-23).  audio_input / audio_output are codec and device I/O and
// stay the reference's own classes; audio_spectrum is new.
#include "infra/processor.hpp"
#include "processor/audio-mix.hpp"
#include "processor/audio-velocity.hpp"
#include "processor/audio-vol.hpp"

namespace infra
{
	namespace
	{
		template <typename... Nodes>
		void register_each()
		{
			(Processor::register_processor<Nodes>(), ...);
		}
	}

	void register_all_processors()
	{
		using namespace processor;
		register_each<Audio_vol, Velocity_modifier, Pitch_modifier, Audio_amix, Audio_bimix, Audio_bimix_v2, Audio_spectrum>();
	}
}
