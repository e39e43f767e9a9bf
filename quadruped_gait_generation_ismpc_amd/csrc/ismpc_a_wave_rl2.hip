// Formulation A wavefront-per-QP kernels, 2 ZMP rows per lane (C <= 128): one translation unit per rows-per-lane value so
// that the instantiations (2 precisions x 4 footstep counts x per-instance yes/no) compile side by side.
#include "ismpc_a_wave.hpp"
namespace ismpc_a { int launch_wave_rl2(const WaveLaunch& L, hipError_t* err) { return launch_wave<2>(L, err); } }
