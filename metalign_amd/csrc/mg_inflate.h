// mg_inflate.h — the device inflater (mg_inflate.hip) as the streaming entry points of mg_stream.hip see it.  Not part of the ABI.
#pragma once
#include <functional>

#include "mg_internal.h"

namespace mg {
// .gz inputs of mg_sketch_stream_add_file / mg_sam_stream_file go through the device (mg_inflate_config's `on`)
bool inflate_dev_enabled();
// The file's text, stage by stage: consume(d_text, nbytes, final, &consumed) as in mg_stream.hip's pipeline — [consumed, nbytes) is
// carried in front of the next stage's text on the device.
int inflate_file_pipeline(int fd, uint64_t fsize, const std::function<int(const uint8_t*, uint64_t, bool, uint64_t*)>& consume,
                          bool* started);  // *started: a piece of text has been handed to `consume` (an error before that: nothing was taken)
}  // namespace mg
