// placeholder until the encoder kernels land (next commit): every entry point fails loudly.
#include "common.hpp"
using namespace kr;
extern "C" {
int kr_encoder_create(const kr_bert_cfg*, int, kr_encoder** out) { if (out) *out = nullptr; return fail(KR_ESTATE, "encoder not built yet"); }
void kr_encoder_destroy(kr_encoder*) {}
int kr_encoder_load_weight(kr_encoder*, const char*, const float*, int64_t) { return fail(KR_ESTATE, "encoder not built yet"); }
int kr_encoder_finalize(kr_encoder*) { return fail(KR_ESTATE, "encoder not built yet"); }
int kr_encoder_forward(kr_encoder*, const int64_t*, const int64_t*, int, int, int, float*, void*) { return fail(KR_ESTATE, "encoder not built yet"); }
int kr_encoder_last_hidden(kr_encoder*, float*, int, int) { return fail(KR_ESTATE, "encoder not built yet"); }
}
