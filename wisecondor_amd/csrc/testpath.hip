// placeholder until the test-path kernels land (replaced in the next commit)
#include "ctx.h"
struct wc_reference { int unused; };
#define NOTYET wc::set_error("test path not built yet"); return WC_E_INTERNAL
extern "C" {
wc_reference *wc_reference_create(wc_ctx *, const int32_t *, const double *, int64_t, int, const int64_t *,
                                  const int64_t *, int, const uint8_t *, const double *, const double *, int, int) {
    wc::set_error("test path not built yet");
    return nullptr;
}
void wc_reference_destroy(wc_reference *) {}
double wc_reference_cutoff(const wc_reference *) { return NAN; }
int wc_optimal_cutoff(wc_ctx *, const double *, int64_t, int, double *) { NOTYET; }
int wc_prepare_samples(wc_ctx *, const wc_reference *, const int32_t *, int64_t, double *, double *) { NOTYET; }
int wc_repeat_test(wc_ctx *, const wc_reference *, const double *, int64_t, double, int, double *, double *,
                   double *, double *) { NOTYET; }
int wc_stouffer_segments(wc_ctx *, const double *, const int64_t *, int64_t, double, int, int, double *,
                         int32_t *, double *, int32_t *, int32_t *) { NOTYET; }
int wc_test_batch(wc_ctx *, const wc_reference *, const int32_t *, int64_t, double, int, int, const int32_t *,
                  int, int, double *, double *, double *, double *, int32_t *, double *) { NOTYET; }
int wc_test_batch_dev(wc_ctx *, void *, const wc_reference *, const int32_t *, int64_t, double, int, int,
                      const int32_t *, int, int, double *, double *, double *, double *, int32_t *, double *) { NOTYET; }
}
