"""AV1 Dr_Intra_Derivative (spec table; reference: dr_intra_derivative[90], EbIntraPrediction.c:299-330),
non-zero entries only; checked against the reference array in test_oracle_vs_ref.py."""
DR_DERIV = {3: 1023, 6: 547, 9: 372, 14: 273, 17: 215, 20: 178, 23: 151, 26: 132, 29: 116, 32: 102, 36: 90, 39: 80,
            42: 71, 45: 64, 48: 57, 51: 51, 54: 45, 58: 40, 61: 35, 64: 31, 67: 27, 70: 23, 73: 19, 76: 15, 81: 11,
            84: 7, 87: 3}
