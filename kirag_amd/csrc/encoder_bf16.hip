// bf16-operand instance of the sentence encoder (namespace kr::enc_bf16): see the header of encoder.hip
#include "encoder.hip"
