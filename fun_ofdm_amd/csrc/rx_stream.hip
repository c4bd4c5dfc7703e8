// rx_stream.hip -- foa_stream_* and foa_shard_* of include/fun_ofdm_amd.h: fun::receiver_chain::process_samples() with everything
// on one device, or dealt batch by batch over several.  Host code only: the kernels are the other units'.
#include "stream_engine.h"
#include "shard_engine.h"
