"""In-kernel phase clocks of the FFN up projection (weights-stationary GEMM with the GELU epilogue): per wave, the time
spent waiting at the tile barrier, in the MFMA phase and in the epilogue.  Needs a library built with -DHRS_ENC_PROF
(hrs_encoder.hip), e.g. hybrid_rag_system_amd/libhrs_encprof.so.
usage: HRS_LIBRARY=$PWD/hybrid_rag_system_amd/libhrs_encprof.so python tools/enc_phases.py"""
import ctypes
import os
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from hybrid_rag_system_amd.engine import Engine
from hybrid_rag_system_amd import weights as W

eng = Engine(0)
cfg = W.EncoderConfig()
enc = eng.encoder(W.generate(cfg), cfg)
ids = torch.from_numpy(np.random.default_rng(0).integers(1000, 30000, (1024, 256)).astype(np.int32)).cuda()
lens = torch.full((1024,), 256, dtype=torch.int32, device='cuda')
for _ in range(2):
    enc.encode_ids(ids, lens)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["HRS_LIBRARY"])
lib.hrs_encoder_prof_dump()  # clears what the warm-up left
enc.encode_ids(ids, lens)
torch.cuda.synchronize()
lib.hrs_encoder_prof_dump()  # six launches (one per layer) summed
