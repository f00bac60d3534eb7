// attention_t.h -- interface between attention.hip (dispatch, C ABI) and attention_t.hip (the transposed-score kernels).
#pragma once
#include <stddef.h>
#include <stdint.h>

struct AttnTArgs {
    const void* qkv;        // [B*T][3*H*dp] bf16
    const void* tab;        // ss_relpos_attention_prepare_tables output
    void* out;              // [B*T][H*dp] bf16 (forward)
    float* lse;             // [B][H][T]
    void* pimg;             // saved probabilities, or null (forward only)
    const void* dO;         // [B*T][H*dp] bf16 (backward)
    const void* O;          // forward output (backward: D = rowsum(dO * O))
    float* Dv;              // [B][H][T] scratch
    void* dqkv;             // [B*T][3*H*dp] bf16
    int B, H, T, dp, D;
    float scale, dropout_p;
    uint64_t seed;
    uint32_t stream_id;
    // x3 (f32 operands as hi / lo bf16 planes): the lo planes of qkv / out / dO / O / dqkv; the table then holds [hi | lo], the image [hi | lo]
    const void* qkv_lo; void* out_lo; const void* dO_lo; const void* O_lo; void* dqkv_lo;
};

bool attn_t_supported(int T, int dp, int D);                    // shape limits and LDS budget of all three kernels
int64_t attn_t_saved_bytes(int B, int H, int T);
int64_t attn_t_table_bytes(int H, int dp);
int attn_t_prepare_tables(const float* emb, int H, int D, int dh, int dp, float scale, void* tab, void* stream);
int attn_t_forward(const AttnTArgs& a, void* stream);
int attn_t_backward(const AttnTArgs& a, void* stream);
// x3: the same kernels' formulation on hi / lo planes (three bf16 MFMAs per product); tables = 2 x attn_t_table_bytes, image = 2 x attn_t_saved_bytes
bool attn_t_x3_supported(int T, int dp, int D);
int attn_t_prepare_tables_x3(const float* emb, int H, int D, int dh, int dp, float scale, void* tab, void* stream);
int attn_t_forward_x3(const AttnTArgs& a, void* stream);
int attn_t_backward_x3(const AttnTArgs& a, void* stream);
