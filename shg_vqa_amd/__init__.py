"""shg_vqa_amd - MI355X (gfx950) implementation of the SHG-VQA hot path.

Host side: PyTorch-ROCm modules with the reference's class names, forward() signatures and
state_dict keys (LXRTEncoder, HGDecoder, CrossEncoder, HungarianMatcher, AGQAModel, BertAdam and the
agqaHGQA training-loop entry points).  Device side: hand-written HIP kernels in libshgvqa.so reached
through the C ABI of include/shg_vqa.h.  There is no CPU or eager fallback: without the built
library (and a GPU) the ops raise.
"""
__version__ = "0.1.0"
