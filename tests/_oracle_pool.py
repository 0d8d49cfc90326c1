"""Session-wide pool of CPU oracle replays for the GPU parity suite (test infrastructure).

The heavy parity tests come in pairs: a `*_device_run` test (marker oracle_submit, collected FIRST) runs the device side,
records the draws and SUBMITS one tests/_oracle_child.py job per slice / seed / precision to this pool; the matching verdict test
(marker oracle_join, collected LAST) JOINS them and asserts.  Between the two the rest of the suite keeps the GPU busy while
the replays run side by side on the host cores -- round 5's suite serialised ~25 independent CPU jobs behind an idle GPU and
did not fit the driver's 1200 s window.

Cores: the physical cores (first half of the affinity list; the SMT siblings are the second half on the GPU boxes) minus a
block reserved for the pytest process itself (in-process oracle comparisons, kernel launches); every job gets its own
disjoint block (unpinned, five 32-thread torch processes ran 7x slower than one alone).  Jobs start in submission order as
blocks free up (first fit), so the longest replay is submitted first.
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


class OraclePool:
    def __init__(self, reserve_main=24, pin_main=True):
        cpus = sorted(os.sched_getaffinity(0))
        phys = cpus[:len(cpus) // 2] if len(cpus) >= 16 else list(cpus)
        if len(phys) >= 4 * reserve_main:
            self.main_cores, self.free = phys[-reserve_main:], phys[:-reserve_main]
            if pin_main:
                try:        # the SMT siblings of the reserved block stay with the main process too
                    os.sched_setaffinity(0, set(self.main_cores) | {c + len(cpus) // 2 for c in self.main_cores if c + len(cpus) // 2 in cpus})
                except OSError:
                    pass
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
        return max(1, len(self.main_cores))

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
        lines = ["oracle pool: %d jobs on %d cores (main process keeps %d); seconds since session start" % (len(self.jobs), self.capacity, len(self.main_cores))]
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
