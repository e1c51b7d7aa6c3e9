"""HIP-event timing of individual kernel launches on the stream they are enqueued on (torch's current
stream, which is the stream every accflow op launches on).  Used by bench.py for the roofline numbers:
events are recorded right before / after the launch of the named op inside the timed region and read
back only after the region's final synchronize."""
import torch


class KernelTimer:
    def __init__(self, names):
        self.names = set(names)
        self.records = []  # (name, start_evt, end_evt, work)

    def wants(self, name):
        return name in self.names

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def end(self, name, start, work, detail=None, work_exec=None):
        """work = algorithmic work of the launch (what the reference's formulation of the op costs), work_exec = what the
        launch really executed when that differs (hoisted / re-indexed convolutions); default: the same."""
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        self.records.append((name, start, e, work, detail, work if work_exec is None else work_exec))

    def by_detail(self, name):
        """per-`detail` breakdown of one op: {detail: dict(launches, total_ms, work)}"""
        out = {}
        for n, s, e, work, detail, wexec in self.records:
            if n != name:
                continue
            d = out.setdefault(detail, dict(launches=0, total_ms=0.0, work=0.0, work_exec=0.0))
            d["launches"] += 1
            d["total_ms"] += s.elapsed_time(e)
            d["work"] += work
            d["work_exec"] += wexec
        return out

    def summary(self):
        """-> {name: dict(launches, total_ms, avg_us, work)}; call after torch.cuda.synchronize()."""
        out = {}
        for name, s, e, work, _, wexec in self.records:
            d = out.setdefault(name, dict(launches=0, total_ms=0.0, work=0.0, work_exec=0.0))
            d["launches"] += 1
            d["total_ms"] += s.elapsed_time(e)
            d["work"] += work
            d["work_exec"] += wexec
        for d in out.values():
            d["avg_us"] = 1e3 * d["total_ms"] / max(1, d["launches"])
        return out


ACTIVE = None  # set by bench.py around the timed region
