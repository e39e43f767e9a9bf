// Formulation A wavefront-per-QP kernels, 3 ZMP rows per lane (C <= 192): one translation unit per rows-per-lane value so
// that the instantiations (2 precisions x 4 footstep counts x per-instance yes/no) compile side by side.
#include "ismpc_a_wave.hpp"
namespace ismpc_a { int launch_wave_rl3(const WaveLaunch& L, hipError_t* err) { return launch_wave<3>(L, err); } }
#ifdef ISMPC_A_PHASES
extern "C" int ismpc_a_debug_phases_rl3(unsigned long long* dst, int reset)
{
    if (dst && hipMemcpyFromSymbol(dst, HIP_SYMBOL(ismpc_a::g_phase), sizeof(unsigned long long) * ismpc_a::NPH) != hipSuccess) return -2;
    if (reset) { unsigned long long z[ismpc_a::NPH] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ismpc_a::g_phase), z, sizeof(z)) != hipSuccess) return -2; }
    return 0;
}
#endif
