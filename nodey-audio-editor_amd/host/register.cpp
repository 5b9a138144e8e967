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

	void register_all_processors(processor::Stretch_algorithm default_algorithm)
	{
		using namespace processor;
		set_default_stretch_algorithm(default_algorithm);
		register_each<Audio_vol, Velocity_modifier, Pitch_modifier, Audio_amix, Audio_bimix, Audio_bimix_v2, Audio_spectrum>();
	}

	// the reference's signature (src/register.cpp:14, called from App::App): the vocoder is this library's default
	void register_all_processors() { register_all_processors(processor::Stretch_algorithm::Vocoder); }
}
