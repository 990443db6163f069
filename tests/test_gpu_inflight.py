"""Several drawings in flight on one GPU (bench.py --inflight: one thread + one stream + one
DrawingPipeline per drawing): the deterministic stages give bit-identical results whatever runs
beside them, and the bench's in-flight loop reports what it ran."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_deterministic_stages_do_not_depend_on_what_runs_beside_them(dev):
    """Contour removal (FFC-ResNet generator + host TELEA) and stage 1 + 2 stylisation are
    deterministic: two pipelines on two threads / streams, each working on its own inputs at the same
    time (process-wide lazily built tables included: the first calls race for them), return exactly
    what the same calls return one after the other."""
    from drawingspinup_amd.drawing import DrawingPipeline, synthetic_drawing, synthetic_edges, synthetic_frames
    pipes = [DrawingPipeline(dev, seed=0, n_frames=4, with_mv=False, with_contour=True) for _ in range(2)]
    inputs = []
    for k in range(2):
        fr = synthetic_frames(40 + k, 4, device=dev)
        inputs.append((synthetic_drawing(40 + k, device=dev), fr, synthetic_edges(fr)))
    torch.cuda.synchronize()
    out = [[None] * 3 for _ in range(2)]
    errors = []

    def work(k, rep):
        try:
            s = torch.cuda.Stream(dev)
            with torch.cuda.stream(s):
                d, fr, ed = inputs[k]
                out[k][rep] = (pipes[k].remove_contour(d).clone(), pipes[k].stylize(fr, ed).clone())
                s.synchronize()
        except BaseException as e:          # noqa: BLE001
            errors.append(e)
    for rep in range(2):                    # rep 0: cold caches, rep 1: warm
        th = [threading.Thread(target=work, args=(k, rep)) for k in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
    assert not errors, errors
    for k in range(2):                      # alone, one after the other
        work(k, 2)
    assert not errors, errors
    for k in range(2):
        for rep in range(2):
            assert torch.equal(out[k][rep][0], out[k][2][0]), (k, rep, "contour")
            assert torch.equal(out[k][rep][1], out[k][2][1]), (k, rep, "frames")


def test_bench_drawing_with_two_drawings_in_flight(dev):
    import bench
    from drawingspinup_amd import dist as ddist
    args = bench.parse(["--config", "drawing", "--steps", "1", "--warmup", "1", "--mv-steps", "2", "--nsr-steps", "48",
                        "--frames", "4", "--no-cpu-baseline", "--inflight", "2", "--inflight-skew", "0.2"])
    timer = bench.KernelTimer()
    timer.install()
    out = bench.bench_drawing(args, ddist, 0, 1, dev, timer)
    c = out["config"]
    assert c["drawings_per_step"] == 2 and c["latency_s"]["drawings"] == 2
    assert abs(out["value"] - 2 / (out["ms_per_step"] * 1e-3)) < 1e-9 * out["value"]
    assert "2 drawings in flight" in c["workload"] and c["inflight_schedule"]["start_skew_s"] == 0.2
    assert c["latency_s"]["max"] <= out["ms_per_step"] * 1e-3 + 1e-6
    r = out["roofline"]
    assert r["in_flight"] == 2 and r["alone"] and r["alone"][0]["launches"] > 0
