"""`python bench.py --gpus N` typed without a launcher: start the N ranks as child processes - decided before anything in the parent
touches torch or HIP (never re-exec a process that initialised the GPU) - relay rank 0's line, check n_gpus."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def spawn_ranks(args):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH] + sys.argv[1:]
    if os.environ.get("LCX_BENCH_DRY_SPAWN"):          # CPU test hook: show the launch, start nothing
        sys.stdout.write(json.dumps({"spawn": cmd}) + "\n")
        return 0
    rc = subprocess.call([sys.executable, os.path.join(ROOT, "__graft_entry__.py")], cwd=ROOT, stdout=sys.stderr)
    if rc != 0:
        sys.stderr.write("bench.py: building the HIP library failed\n")
        return rc
    sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: launching the ranks as children: %s\n" % (args.gpus, " ".join(cmd)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.strip()]
    rec, raw = None, None
    for ln in reversed(lines):
        try:
            rec, raw = json.loads(ln), ln
            break
        except ValueError:
            continue
    if p.returncode != 0 or rec is None:
        sys.stderr.write("bench.py: the rank launch failed (rc %d)\n%s\n" % (p.returncode, "\n".join(lines[-20:])))
        return p.returncode or 1
    if rec.get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: asked for %d GPUs, the ranks report n_gpus=%r\n" % (args.gpus, rec.get("n_gpus")))
        return 1
    sys.stdout.write(raw.strip() + "\n")          # rank 0's compact line, byte for byte
    sys.stdout.flush()
    return 0
