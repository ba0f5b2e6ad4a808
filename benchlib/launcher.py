"""`python bench.py --gpus N` with no torchrun around it starts its own ranks.  Nothing here touches HIP."""
from __future__ import annotations

import os
import socket
import subprocess
import sys

def spawn_ranks(args, script) -> int:
    """N rank processes of `script` (bench.py), rank 0's stdout passed through; a rank that fails takes the others down instead of leaving them
    waiting in a collective for ever.  Returns the worst exit status."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VP8_BENCH_CHILD="1", VP8_BENCH_RDZV_KEY=f"bench-{os.getpid()}-{port}", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    import time
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)     # (rank 0's pipe is drained while everybody is watched)
    reader.start()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad and failed is None:
            failed = bad[0]
            sys.stderr.write(f"bench.py: rank {failed} exited with {procs[failed].returncode}: ending the other ranks\n")
            deadline = time.time() + 10.0       # the others may be on their way out themselves (the same error, a group timeout)
            while time.time() < deadline and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()                    # exactly the processes started here, by handle
        time.sleep(0.05)
    reader.join(timeout=5.0)
    rcs = [p.wait() for p in procs]
    sys.stdout.write((got[0] if got else b"").decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)

