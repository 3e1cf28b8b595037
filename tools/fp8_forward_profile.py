"""W8A8 inference forward of the bench model, alone (for rocprofv3 --kernel-trace --stats; tools/rocprof_summary.py --step-marker embed_assemble).
usage (GPU box): rocprofv3 --kernel-trace --stats -d /tmp/pe_f8 -o t -- python tools/fp8_forward_profile.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    m, args = bench.build_model("7b", dev, 2048)
    g = torch.Generator(device=dev).manual_seed(1)
    img = torch.randn(8, 3, 336, 336, device=dev, generator=g).bfloat16()
    tok = torch.randint(3, args.vocab_size, (8, 512), device=dev, generator=g)
    tok[:, 0] = 1
    m.quantize_decode_weights("fp8", prefill=True)
    for _ in range(steps):
        m.forward_inference(tok, 0, img)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
