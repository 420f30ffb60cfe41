"""`python bench.py --gpus N` typed without a launcher: start the N ranks as child processes - decided before anything in the parent
touches torch or HIP (never re-exec a process that initialised the GPU) - relay rank 0's line, check n_gpus.

The job always ends in ONE JSON line.  A rank set that hangs (first contact with RCCL over xGMI is the step that has never run
on hardware: a hang inside ncclCommInitRank is fatal by RCCL's own contract) or fails is killed as a process group and a FRESH child
set is started on the next transport of the ladder below - never a restart inside a rank, never an exec:

    engine   the library's own RCCL communicator per handle (include/lcx.h lcx_comm_init)          the product path
    hook     the library issues the all-reduces through torch.distributed's RCCL group (LCX_EXCHANGE=hook)
    torch    the host sequences the levels and all-reduces between them (LCX_EXCHANGE=torch)
    gloo     LCX_EXCHANGE=hook over a gloo group (host-staged sums: slow, but it needs nothing of RCCL)   last resort, lean job

Every attempt is recorded on the line (`exchange_attempts`: transport, rc, seconds, reason); if every rung fails the line carries
`value: null`, the attempts and the exit code is non-zero.  Inside a rank first contact is bounded too (linearcorex_amd/comm.py,
LCX_FIRST_CONTACT_TIMEOUT_S): a rank stuck there dumps its stacks and exits 3, which ends the attempt at once instead of at its
wall-clock budget."""
import json
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

# (name, environment of the rank set, lean): lean = headline + same-shard pre-run + CPU baseline only (bench.py LCX_BENCH_LEAN)
LADDER = (("engine", {}, False),
          ("hook", {"LCX_EXCHANGE": "hook"}, False),
          ("torch", {"LCX_EXCHANGE": "torch"}, False),
          ("gloo", {"LCX_EXCHANGE": "hook", "LCX_BENCH_BACKEND": "gloo"}, True))
# Fall-back attempts (every rung after the first) on ONE node also pin the sockets that RCCL's and gloo's bootstraps open to the loopback
# interface unless the caller chose one: the interface picked by default is the one thing a bad first attempt may have tripped over
# that no transport rung changes (a container whose hostname does not resolve), and loopback always carries a single-node rendezvous.
# The data path is untouched (xGMI peer-to-peer / shared memory).
FALLBACK_DEFAULTS = {"NCCL_SOCKET_IFNAME": "lo", "GLOO_SOCKET_IFNAME": "lo"}
ATTEMPT_S = 900.0          # LCX_BENCH_ATTEMPT_S: wall-clock budget of one rank set
TOTAL_S = 1700.0           # LCX_BENCH_TOTAL_S: of the whole ladder (the driver allows a --gpus N job 1 800 s)
FIRST_CONTACT_S = 300      # LCX_FIRST_CONTACT_TIMEOUT_S handed to the ranks unless the caller set one (a rank may enter first contact
                           # minutes before a rank whose first `import torch` on a fresh box is still paging the image in)


_LIVE = []          # (Popen, token) of the rank set / worker that is running right now: what a SIGTERM of this process must take along


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def rank_command(gpus, port, argv):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH] + list(argv)


def ladder_for(env):
    """The rungs this job may take, first to last: (name, environment laid over the caller's, lean).  A caller's own LCX_EXCHANGE /
    LCX_BENCH_BACKEND is the first rung (that is what was asked for) and its backend is kept on the rungs below it (the tests: gloo,
    several ranks on one GPU); rungs that would repeat a setting already tried are dropped.  LCX_BENCH_LADDER=engine,hook,... picks
    rungs by name."""
    asked = env.get("LCX_BENCH_LADDER")
    if asked:
        names = [t.strip() for t in asked.split(",") if t.strip()]
        unknown = [n for n in names if n not in [r[0] for r in LADDER]]
        if unknown:
            raise SystemExit("bench.py: LCX_BENCH_LADDER names unknown rungs %r (known: %s)" % (unknown, ", ".join(r[0] for r in LADDER)))
        return [r for n in names for r in LADDER if r[0] == n]
    ex, bk = env.get("LCX_EXCHANGE"), env.get("LCX_BENCH_BACKEND")
    if not (ex or bk):
        return list(LADDER)
    own = ", ".join("%s=%s" % kv for kv in (("LCX_EXCHANGE", ex), ("LCX_BENCH_BACKEND", bk)) if kv[1])
    rungs, seen = [("caller (%s)" % own, {}, False)], {(ex or "engine", bk or "nccl")}
    for name, extra, lean in LADDER[1:]:
        eff = dict(extra)
        if bk:
            eff["LCX_BENCH_BACKEND"] = bk
        key = (eff.get("LCX_EXCHANGE", ex or "engine"), eff.get("LCX_BENCH_BACKEND", "nccl"))
        if key not in seen:
            seen.add(key)
            rungs.append((name, eff, lean))
    return rungs


def _tagged_pids(token):
    """Processes of this user that carry the attempt's token in their environment: the ranks (torchrun starts each in a session of its
    own, so the launcher's process group does not contain them) and anything they started."""
    needle = ("LCX_BENCH_ATTEMPT_TOKEN=%s" % token).encode()
    found = []
    for name in os.listdir("/proc"):
        if not name.isdigit() or int(name) == os.getpid():
            continue
        try:
            with open("/proc/%s/environ" % name, "rb") as f:
                if needle in f.read().split(b"\0"):
                    found.append(int(name))
        except OSError:
            continue
    return found


def _end_rank_set(p, token, grace=15.0):
    """SIGTERM to the launcher's process group (torchrun forwards it to its ranks), then SIGKILL to it and to every process that still
    carries the attempt's token - exactly the processes this attempt started, found by a token only they have."""
    for sig, wait in ((signal.SIGTERM, grace), (signal.SIGKILL, 5.0)):
        if p.poll() is not None:
            break
        try:
            os.killpg(p.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            p.wait(timeout=wait)
        except subprocess.TimeoutExpired:
            pass
    deadline = time.time() + 10.0
    while True:
        left = _tagged_pids(token)
        if not left or time.time() > deadline:
            return left
        for pid in left:
            try:
                os.kill(pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
        time.sleep(0.2)


def _last_json(text):
    for ln in reversed([t for t in text.splitlines() if t.strip()]):
        try:
            rec = json.loads(ln)
            if isinstance(rec, dict):
                return rec
        except ValueError:
            continue
    return None


def run_attempt(cmd, env, budget_s):
    """One rank set, launcher in a process group of its own.  -> (rc, or None when it was killed at its budget; seconds; stdout text)"""
    t0 = time.time()
    token = "%d-%d" % (os.getpid(), time.time_ns())
    env = dict(env, LCX_BENCH_ATTEMPT_TOKEN=token)
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, start_new_session=True)
    _LIVE[:] = [(p, token)]
    try:
        out, _ = p.communicate(timeout=budget_s)
        rc = p.returncode
    except subprocess.TimeoutExpired:
        rc = None
        _end_rank_set(p, token)
        try:
            out, _ = p.communicate(timeout=5)
        except Exception:          # noqa: BLE001 - the pipe of a killed child: whatever is there
            out = b""
    if rc != 0:
        left = _end_rank_set(p, token, grace=2.0)          # a failed torchrun may leave ranks behind: none may meet the next rank set
        if left:
            sys.stderr.write("bench.py: processes of the failed attempt still alive: %r\n" % (left,))
    _LIVE[:] = []
    return rc, time.time() - t0, (out or b"").decode(errors="replace")


# ------------------------------------------------------------------------------------------------------------------------------------
# The same ladder when somebody ELSE launched the ranks (the driver's `python -m torch.distributed.run ... bench.py --gpus N`): then
# every rank process is a supervisor - it never imports torch, never touches the GPU - that runs the real rank as a child
# (LCX_BENCH_WORKER=1) on a rendezvous of the attempt's own, and the supervisors of one node agree through files in /tmp:
#     ready, built      rank 0: its pid (names the job's directory), the build's rc
#     <k>.port          rank 0: the attempt's rendezvous port          <k>.fail          whoever saw its child fail / exceed the budget
#     <k>.ok.<rank>     this rank's child finished with rc 0           <k>.done.<rank>   this rank's child is gone (before attempt k + 1 starts)
# A rank set hung in first contact therefore costs one attempt, not the job: all children are killed, a fresh set starts on the next rung.
# ------------------------------------------------------------------------------------------------------------------------------------
def _write_atomic(path, text):
    tmp = "%s.tmp.%d" % (path, os.getpid())
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def _wait_for(paths, limit_s, stop=None):
    """until every path exists (True), `stop` exists (False), or limit_s passed (None)"""
    t_end = time.time() + limit_s
    while time.time() < t_end:
        if stop and os.path.exists(stop):
            return False
        if all(os.path.exists(q) for q in paths):
            return True
        time.sleep(0.05)
    return None


def supervise_rank(args, argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    env0 = dict(os.environ)
    rank, world = int(env0.get("RANK", "0")), int(env0.get("WORLD_SIZE", "1"))
    rungs = ladder_for(env0)
    attempt_s = float(env0.get("LCX_BENCH_ATTEMPT_S", ATTEMPT_S))
    total_s = float(env0.get("LCX_BENCH_TOTAL_S", TOTAL_S))
    # the job's coordination directory: /tmp/lcx_sup_<MASTER_PORT>/<pid of rank 0's supervisor>.  The other ranks take the pid from
    # `ready` and believe it only while that process is alive and IS this job's rank-0 supervisor (a stale file of an earlier job on the
    # same port names a dead or foreign process) - no assumption about who the ranks' parent is
    root = os.path.join("/tmp", "lcx_sup_%s" % env0.get("MASTER_PORT", "0"))
    if rank == 0:
        box = os.path.join(root, str(os.getpid()))
        os.makedirs(box, exist_ok=True)
        _write_atomic(os.path.join(root, "ready"), "%d\n" % os.getpid())
        rc = subprocess.call([sys.executable, os.path.join(ROOT, "__graft_entry__.py")], cwd=ROOT, stdout=sys.stderr)      # build once, before any rank
        _write_atomic(os.path.join(box, "built"), "%d\n" % rc)
        for name in os.listdir(root):                                     # directories of earlier jobs on this port
            old = os.path.join(root, name)
            if name.isdigit() and name != str(os.getpid()) and not os.path.exists("/proc/%s" % name):
                for q in os.listdir(old):
                    os.remove(os.path.join(old, q))
                os.rmdir(old)
    else:
        box, t_end = None, time.time() + 900.0
        while box is None and time.time() < t_end:
            try:
                with open(os.path.join(root, "ready")) as f:
                    pid0 = int(f.read().strip())
                with open("/proc/%d/environ" % pid0, "rb") as f:
                    env_of = dict(kv.split(b"=", 1) for kv in f.read().split(b"\0") if b"=" in kv)
                if (env_of.get(b"RANK") == b"0" and env_of.get(b"MASTER_PORT", b"").decode() == env0.get("MASTER_PORT", "0")
                        and env_of.get(b"WORLD_SIZE", b"").decode() == env0.get("WORLD_SIZE") and b"LCX_BENCH_WORKER" not in env_of):
                    box = os.path.join(root, str(pid0))
            except (OSError, ValueError):
                pass
            if box is None:
                time.sleep(0.1)
        if box is None:
            sys.stderr.write("bench.py: rank %d: rank 0's supervisor never appeared (%s)\n" % (rank, root))
            return 1
    if _wait_for([os.path.join(box, "built")], 900.0) is not True:
        sys.stderr.write("bench.py: rank %d: rank 0's supervisor never finished the build (%s)\n" % (rank, box))
        return 1
    with open(os.path.join(box, "built")) as f:
        if int(f.read().strip() or 1) != 0:
            sys.stderr.write("bench.py: rank %d: building the HIP library failed\n" % rank)
            return 1
    worker = json.loads(env0["LCX_BENCH_WORKER_CMD"]) if env0.get("LCX_BENCH_WORKER_CMD") else [sys.executable, BENCH]      # (test hook)
    t_job = time.time()
    attempts, rec = [], None
    if rank == 0:
        _LineOnSigterm(args, attempts).__enter__()          # for the rest of this process's life
    for k, (name, extra, lean) in enumerate(rungs):
        f_port, f_fail = os.path.join(box, "%d.port" % k), os.path.join(box, "%d.fail" % k)
        if rank == 0:
            # rank 0's clock decides for everybody - whether the attempt starts, its budget, whether it is a lean job: the workers of one
            # attempt must run the same program (a lean rank beside a full one would meet in different collectives)
            left = total_s - (time.time() - t_job)
            plan = {"skip": bool(left < 60 and attempts), "left": left, "port": _free_port()}
            plan["budget"] = max(min(30.0, attempt_s), min(attempt_s, left - 30.0))
            plan["lean"] = bool(lean or (k > 0 and plan["budget"] < 0.5 * attempt_s))
            _write_atomic(f_port, json.dumps(plan) + "\n")
        if _wait_for([f_port], 120.0) is not True:
            attempts.append({"transport": name, "rc": None, "seconds": 0.0, "reason": "rank 0's supervisor did not open the attempt"})
            break
        with open(f_port) as f:
            plan = json.loads(f.read())
        if plan["skip"]:
            attempts.append({"transport": name, "rc": None, "seconds": 0.0, "reason": "not started: %.0f s of the job's %.0f s left" % (plan["left"], total_s)})
            continue
        budget, port = float(plan["budget"]), str(plan["port"])
        env = dict(env0, LCX_BENCH_WORKER="1", MASTER_PORT=port, TORCHELASTIC_USE_AGENT_STORE="False")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # (see spawn_ranks)
        env.setdefault("NCCL_DEBUG", "WARN")
        env.setdefault("LCX_FIRST_CONTACT_TIMEOUT_S", str(FIRST_CONTACT_S))
        env.setdefault("LCX_BENCH_LINE_RESERVE", "600")
        env.update(extra)
        if k > 0:
            for key, val in FALLBACK_DEFAULTS.items():
                env.setdefault(key, val)
        if plan["lean"]:
            env["LCX_BENCH_LEAN"] = "1"
        token = "%d-%d" % (os.getpid(), time.time_ns())
        env.update(LCX_BENCH_ATTEMPT="%d:%s" % (k + 1, name), LCX_BENCH_ATTEMPT_TOKEN=token)
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d under a launcher, attempt %d (%s, <= %.0f s%s): every rank supervises a worker child\n"
                             % (world, k + 1, name, budget, ", lean" if env.get("LCX_BENCH_LEAN") else ""))
        t0 = time.time()
        p = subprocess.Popen(worker + argv, cwd=ROOT, env=env, stdout=subprocess.PIPE)
        _LIVE[:] = [(p, token)]
        chunks, reason, rc = [], None, None
        os.set_blocking(p.stdout.fileno(), False)
        while True:
            try:
                data = p.stdout.read()
                if data:
                    chunks.append(data)
            except (BlockingIOError, OSError):
                pass
            rc = p.poll()
            if rc is not None:
                reason = "ok" if rc == 0 else "rank %d's worker exited with rc %d" % (rank, rc)
                break
            if os.path.exists(f_fail):
                reason = "another rank's worker failed"
                break
            if time.time() - t0 > budget:
                reason = "rank %d's worker killed after the attempt's wall-clock budget of %.0f s" % (rank, budget)
                break
            time.sleep(0.1)
        if reason != "ok":
            if not os.path.exists(f_fail):
                try:
                    _write_atomic(f_fail, reason + "\n")
                except OSError:
                    pass
            if p.poll() is None:
                p.terminate()
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            for pid in _tagged_pids(token):
                try:
                    os.kill(pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
        try:
            rest = p.stdout.read()
            if rest:
                chunks.append(rest)
        except (BlockingIOError, OSError, ValueError):
            pass
        out = b"".join(chunks).decode(errors="replace")
        if reason == "ok":
            _write_atomic(os.path.join(box, "%d.ok.%d" % (k, rank)), "ok\n")
            # the attempt counts only if EVERY rank's worker came through
            done = _wait_for([os.path.join(box, "%d.ok.%d" % (k, r)) for r in range(world)], max(60.0, budget - (time.time() - t0)), stop=f_fail)
            if done is not True:
                reason = "another rank's worker failed" if done is False else "the other ranks' workers did not finish"
        if reason != "ok":
            with open(f_fail) as f:
                first = f.read().strip()
            reason = reason if first == reason else "%s (first failure: %s)" % (reason, first)
        attempts.append({"transport": name, "rc": rc, "seconds": round(time.time() - t0, 1), "reason": reason})
        # nobody opens the next attempt while a worker of this one may still hold its GPU
        _write_atomic(os.path.join(box, "%d.done.%d" % (k, rank)), "done\n")
        _wait_for([os.path.join(box, "%d.done.%d" % (k, r)) for r in range(world)], 120.0)
        if reason == "ok":
            rec = _last_json(out) if rank == 0 else {}
            if rank == 0 and (rec is None or rec.get("n_gpus") != args.gpus):
                attempts[-1]["reason"] = reason = "no JSON line on rank 0's stdout" if rec is None else "the ranks report n_gpus=%r" % (rec.get("n_gpus"),)
                rec = None
                # (the other ranks cannot learn of this any more: the job ends here, with the line below)
            break
        if rank == 0:
            sys.stderr.write("bench.py: attempt %d (%s) failed: %s\n" % (k + 1, name, reason))
    ok = bool(attempts) and attempts[-1]["reason"] == "ok"
    f_final = os.path.join(box, "final")
    if rank == 0:
        _emit_line(rec, attempts, args)
        _write_atomic(f_final, "%d\n" % (0 if ok else 1))
    else:
        # a launcher ends the whole job the moment ONE rank exits non-zero: nobody leaves before rank 0 has printed the line
        _wait_for([f_final], 120.0)
    return 0 if ok else 1


def _emit_line(rec, attempts, args, note=None):
    """the one stdout line of a launcher-level job: the worker's line with the attempts, or - nothing finished - a line that says so"""
    if rec is None:
        rec = {"metric": "corex_fit_iterations_per_sec", "value": None, "unit": "fit iterations/s", "n_gpus": args.gpus, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic", "config": {"workload": "c4shard x %d GPUs: no rank set finished" % args.gpus},
               "error": note or "every transport of the ladder failed; see exchange_attempts and stderr"}
    rec["exchange_attempts"] = attempts
    sys.stdout.write(json.dumps(rec, separators=(",", ":")) + "\n")
    sys.stdout.flush()


class _LineOnSigterm:
    """Whoever ends the job from outside (a driver's timeout sends SIGTERM before SIGKILL) still gets the line: the attempts so far,
    `value: null`, and the live child ended with the job."""

    def __init__(self, args, attempts):
        self.args, self.attempts = args, attempts

    def __enter__(self):
        def handler(signum, frame):
            for p, token in list(_LIVE):
                try:
                    os.killpg(p.pid, signal.SIGKILL) if os.getpgid(p.pid) == p.pid else p.kill()
                except Exception:          # noqa: BLE001
                    pass
                for pid in _tagged_pids(token):
                    try:
                        os.kill(pid, signal.SIGKILL)
                    except Exception:          # noqa: BLE001
                        pass
            _emit_line(None, self.attempts + [{"transport": "-", "rc": None, "seconds": 0.0, "reason": "the job was terminated by signal %d" % signum}],
                       self.args, note="terminated from outside before a rank set finished; see exchange_attempts")
            os._exit(1)
        self.old = signal.signal(signal.SIGTERM, handler)
        return self

    def __exit__(self, *exc):
        signal.signal(signal.SIGTERM, self.old)
        return False


def spawn_ranks(args, argv=None, runner=run_attempt):
    argv = list(sys.argv[1:] if argv is None else argv)
    base_env = dict(os.environ)
    rungs = ladder_for(base_env)
    attempt_s = float(base_env.get("LCX_BENCH_ATTEMPT_S", ATTEMPT_S))
    total_s = float(base_env.get("LCX_BENCH_TOTAL_S", TOTAL_S))
    dry = bool(base_env.get("LCX_BENCH_DRY_SPAWN"))
    if dry:          # CPU test hook: show the launch (first rung) and the ladder, start nothing
        sys.stdout.write(json.dumps({"spawn": rank_command(args.gpus, _free_port(), argv),
                                     "ladder": [{"transport": n, "env": e, "lean": lean} for n, e, lean in rungs],
                                     "attempt_s": attempt_s, "total_s": total_s}) + "\n")
        return 0
    rc = subprocess.call([sys.executable, os.path.join(ROOT, "__graft_entry__.py")], cwd=ROOT, stdout=sys.stderr)
    if rc != 0:
        sys.stderr.write("bench.py: building the HIP library failed\n")
        return rc
    t_job = time.time()
    attempts, rec = [], None
    if runner is run_attempt:
        _LineOnSigterm(args, attempts).__enter__()          # for the rest of this process's life
    for k, (name, extra, lean) in enumerate(rungs):
        left = total_s - (time.time() - t_job)
        if left < 60 and attempts:
            attempts.append({"transport": name, "rc": None, "seconds": 0.0, "reason": "not started: %.0f s of the job's %.0f s left" % (left, total_s)})
            continue
        budget = max(min(30.0, attempt_s), min(attempt_s, left - 30.0))
        env = dict(base_env)
        # The pool's host driver supports dmabuf IPC only: with the legacy mode hipIpcGetMemHandle fails ("invalid argument") and with it
        # RCCL's intra-node P2P set-up and any device-tensor sharing between the ranks.  The GPU boxes export this already; the launcher
        # only makes sure a caller's scrubbed environment does not lose it.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("NCCL_DEBUG", "WARN")               # the first RCCL run on xGMI: its warnings belong in the record's stderr
        env.setdefault("LCX_FIRST_CONTACT_TIMEOUT_S", str(FIRST_CONTACT_S))
        env.update(extra)
        if k > 0:
            for key, val in FALLBACK_DEFAULTS.items():
                env.setdefault(key, val)
        if lean or (k > 0 and budget < 0.5 * attempt_s):
            env["LCX_BENCH_LEAN"] = "1"                     # what is left of the job's time does not fit the riders
        env["LCX_BENCH_ATTEMPT"] = "%d:%s" % (k + 1, name)
        env["LCX_BENCH_WORKER"] = "1"                       # these ranks ARE the workers: this function is their ladder (see supervise_rank)
        env.setdefault("LCX_BENCH_LINE_RESERVE", "600")     # room on the 4 KB line for the attempts record added below
        cmd = rank_command(args.gpus, _free_port(), argv)
        sys.stderr.write("bench.py: --gpus %d, attempt %d (%s, <= %.0f s%s): %s\n"
                         % (args.gpus, k + 1, name, budget, ", lean" if env.get("LCX_BENCH_LEAN") else "", " ".join(cmd)))
        sys.stderr.flush()
        rc, secs, out = runner(cmd, env, budget)
        got = _last_json(out)
        if rc is None:
            reason = "killed after its wall-clock budget of %.0f s" % budget
        elif rc != 0:
            reason = "rank set exited with rc %d" % rc
        elif got is None:
            reason = "no JSON line on stdout"
        elif got.get("n_gpus") != args.gpus:
            reason = "the ranks report n_gpus=%r" % (got.get("n_gpus"),)
        else:
            reason = "ok"
        attempts.append({"transport": name, "rc": rc, "seconds": round(secs, 1), "reason": reason})
        if reason == "ok":
            rec = got
            break
        sys.stderr.write("bench.py: attempt %d (%s) failed: %s\n%s\n" % (k + 1, name, reason, "\n".join(out.splitlines()[-20:])))
    _emit_line(rec, attempts, args)
    return 0 if attempts and attempts[-1]["reason"] == "ok" else 1
