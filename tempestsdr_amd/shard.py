"""Multi-GPU sharding of the autocorrelation sweep.

Successive capture windows are independent; the reference only forms their
running mean (accummulate, frameratedetector.c:51-60).  With G ranks (one
process per GPU) rank g takes windows g, g+G, g+2G, ... of the stream, keeps
per-lag SUMS of |R| locally (tsdrgpu_autocorr_run mode 1), and one all-reduce
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests) of the
frame-lag + line-lag sums makes the global plots: sum / total window count.
The mean of sums differs from the running mean only by f64 rounding.

The frame path does not shard in time (IIR state, autogain, sync detector and
resampler phase are frame-to-frame recurrences): ranks run independent
replicas on their own streams.
"""
import numpy as np


def windows_for_rank(total_windows, rank, world):
    """Window indices rank `rank` processes: round-robin keeps every rank busy
    from the first window on."""
    return list(range(rank, total_windows, world))


class DeviceView:
    """Zero-copy torch view of library-owned device memory (via __cuda_array_interface__)."""

    def __init__(self, ptr, count, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def as_tensor(ptr, count, device):
    import torch
    return torch.as_tensor(DeviceView(ptr, count), device=device)


def allreduce_plots(sums, local_windows, dist=None, mean=True, total_windows=None, force=False):
    """sums: 1-D float64 torch tensor (device or CPU) or numpy array with this
    rank's per-lag sums of |R| (frame lags then line lags).  Returns (plots as
    the same type, total window count); plots are the global means, or the
    global sums when mean=False (the caller then divides on the device,
    tsdrgpu_autocorr_finalize_sums).  `dist` = torch.distributed or None for a
    single process.  total_windows: the global window count when the caller knows
    it (skips the second all-reduce and its host synchronisation).  force: issue the
    collective even in a one-rank group (exercises the RCCL launch path on one GPU)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        total = int(local_windows)
        return ((sums / max(total, 1)) if mean else sums), total
    import torch
    t = sums if isinstance(sums, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(sums, np.float64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)    # per-lag energies of every rank's windows
    if total_windows is None:                   # ranks may hold different numbers of windows
        cnt = torch.tensor([float(local_windows)], dtype=torch.float64, device=t.device)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total = int(round(float(cnt.item())))   # (host sync; pass total_windows to avoid it)
    else:
        total = int(total_windows)
    out = (t / max(total, 1)) if mean else t
    return (out if isinstance(sums, torch.Tensor) else out.numpy()), total


def detect_mode(frame_plot, line_plot, frame_lo, line_lo, samplerate):
    """lag -> frame rate / height, as the Java GUI derives them from the plots
    (PlotVisualizer.java:200-247 argmax with the first maximum winning;
    Main.java:1301-1303 fps = fs/(offset+idx), :1346-1350 height =
    round(frame_lag/line_lag))."""
    fi = int(np.argmax(frame_plot))
    li = int(np.argmax(line_plot))
    flag, llag = frame_lo + fi, line_lo + li
    return {"frame_lag": flag, "line_lag": llag, "framerate": samplerate / flag,
            "height": int(round(flag / llag)), "linerate": samplerate / llag}
