/* ref_exact_sse.h -- force-included when the reference's libhog/gradientMex.cpp is compiled a SECOND time as its "exact math" flavour:
 * the two approximate SSE instructions behind libhog/sse.hpp:40-41 (RCP = _mm_rcp_ps, RCPSQRT = _mm_rsqrt_ps) are mapped to the
 * correctly rounded 1 / x and 1 / sqrt(x).  No reference source is modified or copied; the fixture fhog_cases_exact.npz it produces pins
 * the MOT_FHOG_EXACT mode of the oracle and of the device (fixture list item 1: "both flavours"). */
#include <xmmintrin.h>
#define _mm_rcp_ps(x) _mm_div_ps(_mm_set1_ps(1.0f), (x))
#define _mm_rsqrt_ps(x) _mm_div_ps(_mm_set1_ps(1.0f), _mm_sqrt_ps(x))
