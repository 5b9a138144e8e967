// register.cpp — mirror of /root/reference/src/register.cpp:14-24 for the nodes on the GPU path.
// Audio_input / Audio_output (FFmpeg demux, SDL preview, LAME export) are codec/device I/O and stay the
// reference's own; in the editor they remain registered from the reference's translation units.
#include "infra/processor.hpp"
#include "processor/audio-mix.hpp"
#include "processor/audio-velocity.hpp"
#include "processor/audio-vol.hpp"

namespace infra
{
	void register_all_processors()
	{
		Processor::register_processor<processor::Audio_vol>();
		Processor::register_processor<processor::Velocity_modifier>();
		Processor::register_processor<processor::Pitch_modifier>();
		Processor::register_processor<processor::Audio_amix>();
		Processor::register_processor<processor::Audio_bimix>();
		Processor::register_processor<processor::Audio_bimix_v2>();
		Processor::register_processor<processor::Audio_spectrum>();  // new node (no reference counterpart)
	}
}
