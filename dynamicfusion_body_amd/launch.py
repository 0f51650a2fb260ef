"""One process per GPU without an external launcher.

`python bench.py --gpus N` (the driver's command shape) has to start its own ranks.  The parent
process here never touches the GPU -- no HIP call, no torch.cuda query -- it only starts N fresh
children with the torch.distributed environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT), relays their output and returns the worst exit code.  Nothing is re-exec'd: a
process that has initialised the GPU must never be replaced (that takes the machine down on this
pool), so the children are ordinary subprocesses of a parent that stays GPU-free.
"""
import os
import socket
import subprocess
import sys


def under_launcher():
    """True when this process already is one rank of a launched job (torch.distributed.run or spawn_ranks)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port():
    """A port that was free a moment ago (the socket is closed before the ranks bind it: a small race that a second
    job on the same host could win; the rendezvous then fails loudly and the caller retries)."""
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(argv, n_ranks, timeout=None, extra_env=None, json_only=False, grace=10.0):
    """Run `python argv...` as n_ranks processes (rank r gets RANK = LOCAL_RANK = r) and wait for all of them.
    Rank 0's stdout is passed through (that is where the one JSON line goes); the other ranks' stdout is dropped,
    every rank's stderr is passed through.  Returns the largest exit code.  If one rank fails the others get `grace`
    seconds to leave by themselves, are then terminated and, another `grace` later, killed (they would otherwise wait in
    a collective for ever); after `timeout` seconds everything still running is killed and reaped (124).  json_only: of rank 0's stdout only lines that
    start with "{" go to stdout, the rest (e.g. gloo's connection banner) to stderr -- one clean JSON line for a parser."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    env0 = dict(os.environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0["MASTER_PORT"] = str(free_port())
    env0["WORLD_SIZE"] = str(n_ranks)
    env0["LOCAL_WORLD_SIZE"] = str(n_ranks)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what this pool's driver supports (RCCL needs it)
    env0.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_ranks)))
    if extra_env:
        env0.update(extra_env)
    procs = []
    for r in range(n_ranks):
        env = dict(env0)
        env["RANK"] = env["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env,
                                      stdout=(subprocess.PIPE if json_only else None) if r == 0 else subprocess.DEVNULL,
                                      text=True if (json_only and r == 0) else None))
    relay = None
    if json_only:
        import threading

        def pump(stream):
            for line in stream:
                (sys.stdout if line.startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
        relay = threading.Thread(target=pump, args=(procs[0].stdout,), daemon=True)
        relay.start()
    import time
    t0 = time.time()
    rc = 0
    alive = list(procs)
    first_failure = None                                  # when the first rank left with a non-zero code
    terminated = None                                     # when the survivors were sent SIGTERM

    def reap(ps, how):
        for q in ps:
            try:
                how(q)
            except OSError:
                pass

    while alive:
        for p in list(alive):
            code = p.poll()
            if code is not None:
                alive.remove(p)
                if code != 0:
                    rc = max(rc, code if code > 0 else 1)
                    if first_failure is None:
                        first_failure = time.time()
        now = time.time()
        # after a failure the other ranks get `grace` seconds to notice and leave on their own (bench.py's ranks learn it from
        # the store and rank 0 still prints its line), then SIGTERM, then -- a rank stuck in a driver call ignores that -- SIGKILL;
        # only the exact processes started here are touched
        if first_failure is not None and alive:
            if terminated is None and now - first_failure > grace:
                reap(alive, lambda q: q.terminate())
                terminated = now
            elif terminated is not None and now - terminated > grace:
                reap(alive, lambda q: q.kill())
        if timeout is not None and now - t0 > timeout:
            reap(alive, lambda q: q.kill())
            for q in alive:                               # no zombies
                try:
                    q.wait(timeout=10)
                except Exception:
                    pass
            return 124
        time.sleep(0.05)
    if relay is not None:
        relay.join(timeout=10)
    return rc
