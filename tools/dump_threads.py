#!/usr/bin/env python3
"""dump_threads.py PID [delay_s] — after `delay_s`, list every thread of PID (name, state, wchan) and send SIGUSR1 to each one, so
that a process running with STARK_HIP_DEBUG_BT=1 prints the native stack of every thread (api.hip: bt_handler).  For hangs on the
GPU box, where no debugger is available."""
import ctypes, os, signal, sys, time
pid = int(sys.argv[1]); delay = float(sys.argv[2]) if len(sys.argv) > 2 else 0
time.sleep(delay)
libc = ctypes.CDLL(None, use_errno=True)
try:
    tids = sorted(int(t) for t in os.listdir(f"/proc/{pid}/task"))
except FileNotFoundError:
    sys.exit(0)
for t in tids:
    try:
        comm = open(f"/proc/{pid}/task/{t}/comm").read().strip()
        stat = open(f"/proc/{pid}/task/{t}/stat").read().rsplit(")", 1)[1].split()
        wchan = open(f"/proc/{pid}/task/{t}/wchan").read().strip()
        print(f"tid {t} {comm:16s} state {stat[0]} wchan {wchan} utime {stat[11]} stime {stat[12]}", file=sys.stderr, flush=True)
    except OSError:
        pass
for t in tids:
    libc.syscall(234, pid, t, int(signal.SIGUSR1))      # tgkill
    time.sleep(0.05)
