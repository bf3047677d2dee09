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


def allreduce_plots(sums, local_windows, dist=None, mean=True):
    """sums: 1-D float64 torch tensor (device or CPU) or numpy array with this
    rank's per-lag sums of |R| (frame lags then line lags).  Returns (plots as
    the same type, total window count); plots are the global means, or the
    global sums when mean=False (the caller then divides on the device,
    tsdrgpu_autocorr_finalize_sums).  `dist` = torch.distributed or None for a
    single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        total = int(local_windows)
        return ((sums / max(total, 1)) if mean else sums), total
    import torch
    t = sums if isinstance(sums, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(sums, np.float64))
    cnt = torch.tensor([float(local_windows)], dtype=torch.float64, device=t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)    # per-lag energies of every rank's windows
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)  # ranks may hold different numbers of windows
    total = int(round(float(cnt.item())))
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
