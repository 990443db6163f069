"""Instant-NSR (NeuS + hash grid) reconstruction on the gfx950 kernels.

Host-side mirror of 2_charactor_reconstructor/instant_nsr/{models,systems}: same module
names, config keys and state_dict layouts; the third-party native ops (tiny-cuda-nn,
nerfacc) are replaced by libdsu_hip.so."""
