"""One of several PROCESSES sharing a GPU (tests/test_gpu_procs.py): synchronous C2-sized pca() calls in a loop for `seconds`,
every frame compared with the first one; prints one JSON line {iters, mismatches, sha, recovered, error}."""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    seconds, n, N, k = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    out = {"iters": 0, "mismatches": 0, "sha": None, "recovered": 0, "error": None}
    try:
        import torch
        from vip_amd import backend as B
        from vip_amd.psfsub import pca
        from vip_amd.synth import synth_adi_device
        ct, ang = synth_adi_device(n, N, seed=0)
        first = pca(ct, ang, ncomp=k, verbose=False, check_memory=False).clone()
        torch.cuda.synchronize()
        out["sha"] = hashlib.sha256(first.cpu().numpy().tobytes()).hexdigest()
        # rendezvous: start the timed loop when the go-file appears (the parent creates it once every child has warmed up)
        print("READY", flush=True)
        go = sys.argv[5]
        t0 = time.time()
        while not os.path.exists(go) and time.time() - t0 < 120:
            time.sleep(0.01)
        t0 = time.time()
        while time.time() - t0 < seconds:
            o = pca(ct, ang, ncomp=k, verbose=False, check_memory=False)
            torch.cuda.synchronize()
            out["iters"] += 1
            if not torch.equal(torch.nan_to_num(o, nan=1234.5), torch.nan_to_num(first, nan=1234.5)):
                out["mismatches"] += 1
        ctx = B.get_context()
        st = ctx.lib.vipmi_check_deferred(ctx.handle)
        if st != 0:
            out["error"] = "vipmi_check_deferred -> %d" % st
        out["recovered"] = max(0, ctx.get_option("eigh_recovered"))
    except Exception as e:      # noqa: BLE001
        out["error"] = repr(e)
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
