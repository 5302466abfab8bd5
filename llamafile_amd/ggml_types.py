"""ggml type ids and block geometry for the quantized-matmul path.

Mirrors include/lfamd_blocks.h (which cites the reference evidence per layout;
SURVEY.md §8 a-0).  Numeric ids are upstream's ``enum ggml_type`` — the values
``llamafile_sgemm`` receives in ``Atype/Btype/Ctype``
(/root/reference/llamafile/sgemm.h:23-24).
"""

F32 = 0
F16 = 1
Q4_0 = 2
Q4_1 = 3
Q5_0 = 6
Q5_1 = 7
Q8_0 = 8
Q8_1 = 9
Q2_K = 10
Q3_K = 11
Q4_K = 12
Q5_K = 13
Q6_K = 14
Q8_K = 15
IQ4_XS = 23
I32 = 26
BF16 = 30

NAMES = {
    F32: "F32", F16: "F16", Q4_0: "Q4_0", Q4_1: "Q4_1", Q5_0: "Q5_0", Q5_1: "Q5_1",
    Q8_0: "Q8_0", Q8_1: "Q8_1", Q2_K: "Q2_K", Q3_K: "Q3_K", Q4_K: "Q4_K", Q5_K: "Q5_K",
    Q6_K: "Q6_K", Q8_K: "Q8_K", IQ4_XS: "IQ4_XS", I32: "I32", BF16: "BF16",
}
BY_NAME = {v: k for k, v in NAMES.items()}

# elements per block, bytes per block
BLCK = {
    F32: 1, F16: 1, BF16: 1, I32: 1,
    Q4_0: 32, Q4_1: 32, Q5_0: 32, Q5_1: 32, Q8_0: 32, Q8_1: 32,
    Q2_K: 256, Q3_K: 256, Q4_K: 256, Q5_K: 256, Q6_K: 256, Q8_K: 256, IQ4_XS: 256,
}
TYPE_SIZE = {
    F32: 4, F16: 2, BF16: 2, I32: 4,
    Q4_0: 18, Q4_1: 20, Q5_0: 22, Q5_1: 24, Q8_0: 34, Q8_1: 36,
    Q2_K: 84, Q3_K: 110, Q4_K: 144, Q5_K: 176, Q6_K: 210, Q8_K: 292, IQ4_XS: 136,
}

# type_traits[].vec_dot_type: which activation format a weight type multiplies with
VEC_DOT = {
    Q4_0: Q8_0, Q5_0: Q8_0, Q8_0: Q8_0, Q4_1: Q8_1, Q5_1: Q8_1,
    Q2_K: Q8_K, Q3_K: Q8_K, Q4_K: Q8_K, Q5_K: Q8_K, Q6_K: Q8_K, IQ4_XS: Q8_K,
    F32: F32, F16: F16, BF16: BF16,
}

QUANT_WEIGHT_TYPES = (Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, IQ4_XS)


def row_size(t: int, ne: int) -> int:
    """ggml_row_size(type, ne): bytes of a row of ``ne`` elements."""
    assert ne % BLCK[t] == 0, (NAMES[t], ne)
    return TYPE_SIZE[t] * (ne // BLCK[t])
