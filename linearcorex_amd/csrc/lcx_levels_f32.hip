// lcx_levels_f32.hip - the float32 half of the moment / update levels (levels_typed.hpp)
#define LCX_T float
#define LCX_NS lcx_f32
#include "levels_typed.inc"
