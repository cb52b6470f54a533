// f16-operand instance of the sentence encoder (namespace kr::enc_f16): see the header of encoder.hip
#define KR_ENC_BUILD_F16 1
#include "encoder.hip"
