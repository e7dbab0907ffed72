// svt_hip_frame.hip — the ONE-LAUNCH form of the per-picture encode pass (enc_frame_kernel<PixT, BD, 3>, kernel_frame.h) in a
// translation unit of its own.  Not a matter of taste: compiled next to the other instantiations of the same bodies
// (svt_hip_txfm.hip: the per-size kernels and the class launches) the compiler's inlining budget leaves this kernel at its
// 168-register cap with 20 B (8-bit) / 36 B (10-bit) of scratch per lane - round 2's state; alone it takes 147 / 160 VGPRs and no
// scratch.  tests/test_kernel_resources.py reads the built library and fails if that ever changes.
#include "host_common.h"
#include "kernel_frame.h"

using namespace svtdev;

namespace svthost {
int launch_enc_frame_one(const svtdev::FrameDesc* fd, uint32_t total_wgs, int is_16bit, hipStream_t s) {
    if (is_16bit) hipLaunchKernelGGL((enc_frame_kernel<uint16_t, 10, 3>), dim3(total_wgs), dim3(256), 0, s, *fd);
    else hipLaunchKernelGGL((enc_frame_kernel<uint8_t, 8, 3>), dim3(total_wgs), dim3(256), 0, s, *fd);
    return launch_status("enc_frame");
}
}  // namespace svthost
