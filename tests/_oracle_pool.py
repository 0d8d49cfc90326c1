"""Session-wide pool of CPU oracle replays for the GPU parity suite (test infrastructure).

The heavy parity tests come in pairs: a `*_device_run` test (marker oracle_submit, collected FIRST) runs the device side,
records the draws and SUBMITS one tests/_oracle_child.py job per slice / seed / precision to this pool; the matching verdict test
(marker oracle_join, collected LAST) JOINS them and asserts.  Between the two the rest of the suite keeps the GPU busy while
the replays run side by side on the host cores -- round 5's suite serialised ~25 independent CPU jobs behind an idle GPU and
did not fit the driver's 1200 s window.

Cores.  The GPU boxes show 256 logical CPUs but their cgroup grants SIXTEEN CPUs of time (/sys/fs/cgroup/cpu.max = "1600000
100000", measured round 6: profiles/r06a_box.txt): a hundred runnable threads are throttled to sixteen cores' worth, every job
-- and the pytest process that launches the kernels -- stretches 3-6x, and round 6's first run of this pool (32 jobs, 104
threads) took the headline replay from 300 s to 900 s.  The pool therefore sizes itself by the quota: capacity = quota - 2
threads (the pytest process keeps two), every job pinned to its own physical cores (first half of the affinity list; the SMT
siblings are the second half); jobs start in submission order as threads free up (first fit), so the longest replay is
submitted first.  What fits that budget is the default suite; the full arbiter statistics run with IPDM_PARITY_FULL=1.
"""
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "_oracle_child.py")


class Job:
    def __init__(self, tag, job, out, threads):
        self.tag, self.job, self.out, self.threads = tag, job, out, threads
        self.proc = self.cores = self.t_start = self.t_end = self.rc = None
        self.t_submit = time.time()
        self.log = out + ".log"


def cpu_quota():
    """CPUs of time the cgroup grants (cpu.max / cfs quota), or the affinity count when there is no quota."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            return max(1, min(n, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
            q, per = int(f.read()), int(g.read())
        if q > 0:
            return max(1, min(n, q // per))
    except (OSError, ValueError):
        pass
    return n


def host_threads():
    """Threads for CPU oracle work INSIDE the pytest process: a quarter of the CPU quota (the pool's replays hold the rest)."""
    return max(2, min(32, cpu_quota() // 4))


class OraclePool:
    def __init__(self, reserve_main=2):
        cpus = sorted(os.sched_getaffinity(0))
        phys = cpus[:len(cpus) // 2] if len(cpus) >= 16 else list(cpus)
        self.quota = cpu_quota()
        budget = min(self.quota, len(phys))
        if budget >= 8:
            # one core in every `stride`: the budgeted threads spread over the sockets' core complexes (their L3s and boost headroom)
            stride = max(1, len(phys) // budget)
            spread = phys[::stride][:budget]
            self.main_cores, self.free = spread[budget - reserve_main:], spread[:budget - reserve_main]
            self.shared = False
        else:               # a small host (the CPU container): no reservation, at most two jobs at a time share what there is
            self.main_cores, self.free, self.shared = list(phys), list(phys), True
        self.capacity = len(self.free)
        self.tmpdir = tempfile.mkdtemp(prefix="ipdm_oracle_pool_")
        self.jobs, self.queue, self.stash = [], [], {}
        self.lock = threading.Lock()
        self.t0 = time.time()
        self._stop = False
        self._pump = threading.Thread(target=self._run, daemon=True)
        self._pump.start()

    # ------------------------------------------------------------------ public
    def main_threads(self):
        """Thread count for oracle work done inside the pytest process."""
        return host_threads()

    def path(self, name):
        return os.path.join(self.tmpdir, name)

    def submit(self, tag, job_npz, threads):
        """job_npz: a file written by tests/_oracle_child.write_job.  Returns the handle to pass to result()."""
        j = Job(tag, job_npz, job_npz[:-4] + ".out.npy", max(1, min(threads, self.capacity)))
        with self.lock:
            self.jobs.append(j)
            self.queue.append(j)
        self._schedule()
        return j

    def result(self, j, timeout=1000.0, mid=False):
        """Blocks until the job has finished; returns its output array (and, with mid, the stored iterates)."""
        import numpy as np
        t_end = time.time() + timeout
        while j.rc is None:
            if time.time() > t_end:
                raise TimeoutError("oracle replay %s not finished after %.0f s (queued %.0f s, running %.0f s)\n%s" % (
                    j.tag, timeout, (j.t_start or time.time()) - j.t_submit, time.time() - (j.t_start or time.time()), self._tail(j)))
            time.sleep(0.2)
        assert j.rc == 0, "oracle replay %s failed (rc %s)\n%s" % (j.tag, j.rc, self._tail(j))
        out = np.load(j.out)
        return (out, np.load(j.out + ".mid.npz")) if mid else out

    def report(self):
        lines = ["oracle pool: %d jobs on %d threads (cgroup quota %d CPUs of %d visible; the pytest process keeps %d); seconds since session start" % (
            len(self.jobs), self.capacity, self.quota, len(os.sched_getaffinity(0)), len(self.main_cores))]
        for j in self.jobs:
            lines.append("%-28s threads %2d submit %6.1f start %6.1f end %6.1f run %6.1f rc %s" % (
                j.tag, j.threads, j.t_submit - self.t0, (j.t_start or 0) - self.t0, (j.t_end or 0) - self.t0,
                (j.t_end or 0) - (j.t_start or 0), j.rc))
        return "\n".join(lines)

    def close(self):
        self._stop = True
        with self.lock:
            for j in self.jobs:
                if j.proc is not None and j.rc is None:
                    j.proc.kill()      # the exact children this pool started
        import shutil
        shutil.rmtree(self.tmpdir, ignore_errors=True)

    # ------------------------------------------------------------------ internals
    def _tail(self, j):
        try:
            return open(j.log).read()[-3000:]
        except OSError:
            return "(no log)"

    def _schedule(self):
        with self.lock:
            for j in list(self.jobs):
                if j.proc is not None and j.rc is None and j.proc.poll() is not None:
                    j.t_end = time.time()
                    if not self.shared:
                        self.free = sorted(self.free + j.cores)
                    j.rc = j.proc.returncode
            for j in list(self.queue):
                running = sum(1 for k in self.jobs if k.proc is not None and k.rc is None)
                if (self.shared and running < 2) or (not self.shared and len(self.free) >= j.threads):
                    j.cores = list(self.free[:j.threads])
                    if not self.shared:
                        self.free = self.free[j.threads:]
                    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(j.threads),
                               MKL_NUM_THREADS=str(j.threads))
                    j.t_start = time.time()
                    j.proc = subprocess.Popen([sys.executable, CHILD, j.job, j.out, str(j.threads), ",".join(map(str, j.cores))],
                                              env=env, cwd=ROOT, stdout=open(j.log, "w"), stderr=subprocess.STDOUT)
                    self.queue.remove(j)

    def _run(self):
        while not self._stop:
            self._schedule()
            time.sleep(0.25)
