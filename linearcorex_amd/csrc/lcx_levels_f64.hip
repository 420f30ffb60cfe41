// lcx_levels_f64.hip - the float64 half of the moment / update levels (levels_typed.hpp)
#define LCX_T double
#define LCX_NS lcx_f64
#include "levels_typed.inc"
