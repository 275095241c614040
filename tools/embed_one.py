"""One short query through the encoder a few times (for a kernel timeline under rocprofv3)."""
import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd import helper
ef = helper.get_embedding_function()
for _ in range(5):
    ef.embed_tensor(["How do I restart the nginx container on Unraid?"])
torch.cuda.synchronize()
