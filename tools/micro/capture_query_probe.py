"""Probe (MI355X, ROCm 7.2): what does hipEventQuery from a SECOND thread answer for an event that was recorded before a
capture on a stream that has since joined the capture?  (ProcessGroupNCCL's watchdog does exactly this with the events of
the warm-up collectives.)  Prints one line per capture error mode."""
import threading
import time

import torch


def probe(mode, join):
    S = torch.cuda.Stream()
    ev = torch.cuda.Event()
    x = torch.zeros(1024, device="cuda")
    with torch.cuda.stream(S):
        x.add_(1)
        ev.record(S)
    torch.cuda.synchronize()
    C = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    res = {}

    def poll():
        try:
            res["query"] = ev.query()
        except Exception as exc:
            res["error"] = str(exc).splitlines()[0][:120]
    err = None
    try:
        with torch.cuda.graph(g, stream=C, capture_error_mode=mode):
            x.mul_(2)
            if join:
                S.wait_stream(C)
                with torch.cuda.stream(S):
                    x.add_(3)
                C.wait_stream(S)
            t = threading.Thread(target=poll)
            t.start()
            t.join()
            x.add_(5)
    except Exception as exc:
        err = str(exc).splitlines()[0][:120]
    torch.cuda.synchronize()
    print(f"mode={mode:12s} event's stream joined the capture={join}: poller -> {res}; capture -> {'ok' if err is None else err}", flush=True)


if __name__ == "__main__":
    import subprocess
    import sys
    if len(sys.argv) == 3:
        probe(sys.argv[1], sys.argv[2] == "1")
    else:                                           # each case in its own process: a refused call may poison the context
        for mode in ("global", "thread_local", "relaxed"):
            for join in ("0", "1"):
                r = subprocess.run([sys.executable, __file__, mode, join], capture_output=True, text=True)
                print((r.stdout.strip() or r.stderr.strip().splitlines()[-1])[:400], flush=True)
