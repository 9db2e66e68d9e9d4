"""Throughput of the device RLE encoder (esam3_rle_encode): n masks of 1008 x 1008, HIP events around the
six-kernel pipeline.  Algorithmic bytes = n*H*W mask bytes in + 4 bytes per run out.

    python tools/rle_bench.py            # on an MI355X
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from efficientsam3_amd import _lib, synth


def main():
    lib = _lib.load()
    dev = torch.device("cuda")
    blob = torch.from_numpy(synth.rle_test_masks()["blobs_1008"]).to(dev)
    res = []
    for n in (32, 200):
        masks = blob[torch.arange(n, device=dev) % blob.shape[0]].contiguous()
        cap = n * 16384
        counts = torch.empty((cap,), dtype=torch.int32, device=dev)
        offs = torch.empty((n + 1,), dtype=torch.int32, device=dev)
        nbytes = int(lib.esam3_rle_scratch_bytes(n, 1008, 1008, cap))
        scratch = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def run():
            _lib.check(lib.esam3_rle_encode(masks.data_ptr(), n, 1008, 1008, counts.data_ptr(), cap, offs.data_ptr(),
                                            scratch.data_ptr(), nbytes, st), "esam3_rle_encode")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        runs = int(offs[-1].item())
        gb = (n * 1008 * 1008 + 4 * runs) / 1e9
        res.append({"masks": n, "ms": round(ms, 4), "runs": runs, "algorithmic_GBps": round(gb / (ms * 1e-3), 1),
                    "masks_per_s": round(n / (ms * 1e-3))})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
