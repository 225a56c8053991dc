#include "rn_fused.h"
int rn_fused_prepare(rn_handle*, const rn_weights*) {
    rn_set_error("16-bit fused path not built yet");
    return RN_E_INVALID;
}
int rn_fused_forward(rn_handle*, const uint8_t*, const float*, int, float*, int64_t*) {
    rn_set_error("16-bit fused path not built yet");
    return RN_E_INVALID;
}
