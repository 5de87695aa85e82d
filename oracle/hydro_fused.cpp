// ORACLE — TEST INFRASTRUCTURE ONLY.  The fused, vectorised CPU form of the hydro flux evaluation (hydro_fused.hpp), in a translation unit of
// its own so that its loops — and only they — are compiled with -fno-math-errno -fno-trapping-math (oracle/Makefile).
#define ORACLE_FUSED_IMPL 1
#include "hydro_fused.hpp"
