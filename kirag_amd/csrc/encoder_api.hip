// C entry points of the sentence encoder (include/kirag_amd.h).  encoder.hip is compiled once per 16-bit operand type (kr::enc_bf16, kr::enc_f16);
// a handle remembers which one it was created with and every call is forwarded to that instance.
#include "common.hpp"

#include <cstdlib>
#include <cstring>

namespace kr {
#define KR_ENC_DECL(NS)                                                                                                                      \
    namespace NS {                                                                                                                           \
    int enc_create(const kr_bert_cfg* cfg, int device, int residual_lo, void** out);                                                         \
    void enc_destroy(void* h);                                                                                                               \
    int enc_load_weight(void* h, const char* hf_name, const float* data, int64_t numel);                                                     \
    int enc_finalize(void* h);                                                                                                               \
    int enc_forward(void* h, const int64_t* input_ids, const int64_t* attention_mask, const int64_t* token_type_ids, int B, int S, int pool, float* out, void* stream); \
    int enc_forward_packed(void* h, const int32_t* token_ids, const int32_t* seq_lens, int B, int S, int64_t total_tokens, int pool, float* out, void* stream); \
    int enc_check(void* h);                                                                                                                  \
    int enc_last_hidden(void* h, float* out, int B, int S);                                                                                  \
    }
KR_ENC_DECL(enc_bf16)
KR_ENC_DECL(enc_f16)

struct EncHandle { int dtype; int residual_lo; void* impl; };
}  // namespace kr

using namespace kr;

#define KR_ENC_CALL(h, fn, ...) ((h)->dtype == KR_ENC_F16 ? enc_f16::fn((h)->impl, ##__VA_ARGS__) : enc_bf16::fn((h)->impl, ##__VA_ARGS__))

extern "C" {

int kr_encoder_create_ex(const kr_bert_cfg* cfg, int device, int operand_dtype, int residual_lo, kr_encoder** out) {
    if (!out || !cfg) return fail(KR_EINVAL, "NULL argument");
    *out = nullptr;
    // defaults (measured on the reference-generated goldens with outlier-channel weights): f16 operands + the residual stream's low
    // half.  The environment may override a default (-1), never an explicit argument.
    if (operand_dtype < 0) {
        const char* v = getenv("KIRAG_AMD_ENCODER_DTYPE");
        if (v && std::strcmp(v, "bf16") == 0) operand_dtype = KR_ENC_BF16;
        else if (v && std::strcmp(v, "f16") == 0) operand_dtype = KR_ENC_F16;
        else if (v && *v) return fail(KR_EINVAL, "KIRAG_AMD_ENCODER_DTYPE='%s': expected bf16 or f16", v);
        else operand_dtype = KR_ENC_DEFAULT_DTYPE;
    }
    if (residual_lo < 0) {
        const char* v = getenv("KIRAG_AMD_RESIDUAL_LO");
        residual_lo = (v && *v) ? (atoi(v) != 0) : KR_ENC_DEFAULT_RESIDUAL_LO;
    }
    if (operand_dtype != KR_ENC_BF16 && operand_dtype != KR_ENC_F16) return fail(KR_EINVAL, "operand_dtype must be KR_ENC_BF16 (0) or KR_ENC_F16 (1)");
    void* impl = nullptr;
    const int rc = operand_dtype == KR_ENC_F16 ? enc_f16::enc_create(cfg, device, residual_lo, &impl) : enc_bf16::enc_create(cfg, device, residual_lo, &impl);
    if (rc) return rc;
    EncHandle* h = new EncHandle{operand_dtype, residual_lo != 0, impl};
    *out = reinterpret_cast<kr_encoder*>(h);
    return 0;
}

int kr_encoder_create(const kr_bert_cfg* cfg, int device, kr_encoder** out) { return kr_encoder_create_ex(cfg, device, -1, -1, out); }

void kr_encoder_destroy(kr_encoder* e) {
    if (!e) return;
    EncHandle* h = reinterpret_cast<EncHandle*>(e);
    if (h->dtype == KR_ENC_F16) enc_f16::enc_destroy(h->impl); else enc_bf16::enc_destroy(h->impl);
    delete h;
}

int kr_encoder_operand_dtype(const kr_encoder* e) { return e ? reinterpret_cast<const EncHandle*>(e)->dtype : -1; }
int kr_encoder_residual_lo(const kr_encoder* e) { return e ? reinterpret_cast<const EncHandle*>(e)->residual_lo : -1; }

int kr_encoder_load_weight(kr_encoder* e, const char* hf_name, const float* data, int64_t numel) {
    if (!e) return fail(KR_EINVAL, "NULL argument");
    return KR_ENC_CALL(reinterpret_cast<EncHandle*>(e), enc_load_weight, hf_name, data, numel);
}
int kr_encoder_finalize(kr_encoder* e) {
    if (!e) return fail(KR_EINVAL, "NULL argument");
    return KR_ENC_CALL(reinterpret_cast<EncHandle*>(e), enc_finalize);
}
int kr_encoder_forward(kr_encoder* e, const int64_t* input_ids, const int64_t* attention_mask, int B, int S, int pool, float* out, void* stream) {
    if (!e) return fail(KR_EINVAL, "encoder is NULL");
    return KR_ENC_CALL(reinterpret_cast<EncHandle*>(e), enc_forward, input_ids, attention_mask, nullptr, B, S, pool, out, stream);
}
int kr_encoder_forward_tt(kr_encoder* e, const int64_t* input_ids, const int64_t* attention_mask, const int64_t* token_type_ids, int B, int S, int pool, float* out,
                          void* stream) {
    if (!e) return fail(KR_EINVAL, "encoder is NULL");
    return KR_ENC_CALL(reinterpret_cast<EncHandle*>(e), enc_forward, input_ids, attention_mask, token_type_ids, B, S, pool, out, stream);
}
int kr_encoder_forward_packed(kr_encoder* e, const int32_t* token_ids, const int32_t* seq_lens, int B, int S, int64_t total_tokens, int pool, float* out,
                              void* stream) {
    if (!e) return fail(KR_EINVAL, "encoder is NULL");
    return KR_ENC_CALL(reinterpret_cast<EncHandle*>(e), enc_forward_packed, token_ids, seq_lens, B, S, total_tokens, pool, out, stream);
}
int kr_encoder_check(kr_encoder* e) {
    if (!e) return fail(KR_EINVAL, "encoder is NULL");
    return KR_ENC_CALL(reinterpret_cast<EncHandle*>(e), enc_check);
}
int kr_encoder_last_hidden(kr_encoder* e, float* out, int B, int S) {
    if (!e || !out) return fail(KR_EINVAL, "NULL argument");
    return KR_ENC_CALL(reinterpret_cast<EncHandle*>(e), enc_last_hidden, out, B, S);
}

}  // extern "C"
