// ref_hog_probe.cpp -- test infrastructure.  Compiles the reference's
// libhog/gradientMex.cpp (included by path, never copied) and exports C entry
// points for the golden-vector generator.
#include REF_GRADIENTMEX_CPP
#include <string.h>
extern "C" {
void refhog_grad_mag(float* I, float* M, float* O, int h, int w) { gradMag(I, M, O, h, w, 1, true); }
void refhog_grad_hist(float* M, float* O, float* R1, int h, int w) { gradHist(M, O, R1, h, w, 4, 18, -1, true); }
// same call sequence as FHoG::extract (libhog/fhog.h:16-38) with plain malloc
void refhog_extract(float* I, int h, int w, float* H) {
  float* M = (float*)alMalloc((size_t)h * w * 4 * sizeof(float), 16); float* O = M + h * w * 2;
  gradMag(I, M, O, h, w, 1, true);
  memset(H, 0, (size_t)(h / 4) * (w / 4) * 32 * sizeof(float));
  fhog(M, O, H, h, w, 4, 9, -1, 0.2f);
  alFree(M);
}
const float* refhog_acos_table(void) { return acosTable(); }
}
