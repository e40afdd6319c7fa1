"""hipGraph capture of small refiner calls (PosePredictor.forward with <= graph_rows rows, materialize=False): the replayed graph must
return exactly what the eager launches return -- same kernels, same order, same buffers -- for new poses, new labels and a new frame."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FIELDS = ("TCO_input", "TCO_output", "K_crop", "KV_crop", "boxes_rend", "boxes_crop", "tCR", "TCV_O_input")


def _model(object_dataset):
    from megapose6d_amd.load_model import build_pose_model
    from megapose6d_amd.mesh_db import MeshDataBase
    from megapose6d_amd.renderer import Panda3dBatchRenderer
    from tests.support import synthetic as syn

    cfg = syn.make_cfg("refiner")
    sd = syn.make_state_dict("vanilla_resnet34", syn.n_inputs_for(cfg), "pose", 9, seed=21)
    renderer = Panda3dBatchRenderer(object_dataset, n_workers=1, preload_cache=True)
    return build_pose_model(cfg, sd, renderer, MeshDataBase.from_object_ds(object_dataset).batched().cuda())


def _inputs(object_dataset, rows, seed):
    from tests.support import synthetic as syn

    rng = np.random.RandomState(seed)
    labels = [object_dataset[int(i)].label for i in rng.randint(0, len(object_dataset.list_objects), rows)]
    T0 = torch.from_numpy(np.stack([syn.random_pose(rng, (0.45, 0.6), 0.1) for _ in labels])).cuda()
    K = torch.from_numpy(np.repeat(syn.K_EXAMPLE[None], rows, 0)).float().cuda()
    images = torch.rand(2, 3, 480, 640, generator=torch.Generator().manual_seed(seed))
    images = (torch.round(images * 255) / 255).cuda()
    im_ids = torch.from_numpy(rng.randint(0, 2, rows).astype(np.int32)).cuda()
    return images, K, labels, T0, im_ids


def _same(a, b, n_iterations):
    for n in range(1, n_iterations + 1):
        x, y = a[f"iteration={n}"], b[f"iteration={n}"]
        for f in FIELDS:
            assert torch.equal(getattr(x, f).contiguous(), getattr(y, f).contiguous()), (n, f)
        assert torch.equal(x.network_outputs["pose"].contiguous(), y.network_outputs["pose"].contiguous()), n


@pytest.mark.parametrize("rows", [1, 5])
def test_graph_replay_is_bit_identical_to_the_eager_launches(object_dataset, rows):
    model = _model(object_dataset)
    model.graph_rows = 16   # (opt-in: MP_REFINER_GRAPH_ROWS / PosePredictor.graph_rows; off by default, see the timing test below)
    n_it = 3
    runs = []
    for seed in (1, 2, 3, 4):   # call 1 eager (warm-up of the shape), call 2 captures + replays, calls 3 / 4 replay with new poses / labels / frame
        images, K, labels, T0, im_ids = _inputs(object_dataset, rows, seed)
        out = model(images=images, K=K, labels=labels, TCO=T0, n_iterations=n_it, im_ids=im_ids, materialize=False)
        runs.append((seed, out))
    assert len(model._graphs) == 1
    model.graph_rows = 0   # eager reference
    for seed, out in runs:
        images, K, labels, T0, im_ids = _inputs(object_dataset, rows, seed)
        ref = model(images=images, K=K, labels=labels, TCO=T0, n_iterations=n_it, im_ids=im_ids, materialize=False)
        _same(out, ref, n_it)
    # results of an earlier replay are the caller's own memory: a later call must not have touched them
    images, K, labels, T0, im_ids = _inputs(object_dataset, rows, 2)
    ref2 = model(images=images, K=K, labels=labels, TCO=T0, n_iterations=n_it, im_ids=im_ids, materialize=False)
    _same(runs[1][1], ref2, n_it)


def test_replays_survive_larger_eager_calls_and_other_row_counts_in_between(object_dataset):
    """A captured call holds raw device addresses.  The eager path's per-slot buffers are grow-only, so a larger eager call (or the warm-up
    of another row count) on the same slot reallocates them: the captured calls therefore run on private buffers, and a replay after such
    calls must still equal the eager result bit for bit."""
    model = _model(object_dataset)
    model.graph_rows = 8
    n_it = 2

    def call(rows, seed):
        images, K, labels, T0, im_ids = _inputs(object_dataset, rows, seed)
        return model(images=images, K=K, labels=labels, TCO=T0, n_iterations=n_it, im_ids=im_ids, materialize=False)

    call(2, 1); call(2, 2)               # warm-up + capture of the 2-row call
    assert len(model._graphs) == 1
    model.graph_rows = 0
    call(40, 3)                          # a larger eager call on the same slot: the slot's CNN-input / workspaces are reallocated
    torch.cuda.empty_cache()             # ... and the old storage really leaves the process
    model.graph_rows = 8
    call(5, 4); call(5, 5)               # warm-up + capture of a second row count
    assert len(model._graphs) == 2
    got2, got5 = call(2, 6), call(5, 7)  # replays of both graphs
    junk = torch.full((64 << 20,), float("nan"), device="cuda")   # whatever memory was freed above is now NaN
    got2b = call(2, 6)
    del junk
    model.graph_rows = 0
    _same(got2, call(2, 6), n_it)
    _same(got2b, call(2, 6), n_it)
    _same(got5, call(5, 7), n_it)
    model.invalidate_graphs()
    assert not model._graphs and not any(isinstance(k, tuple) for k in model._x)   # the private buffers are released with the graphs


def test_graph_replay_timing_for_one_row(object_dataset):
    """Why the capture is off by default: the replay is not faster than the eager launches (measured 5.98 vs 5.96 ms) -- the call is
    bound by the device time of its dependent small-grid kernels.  The test reports both and only guards against a regression."""
    model = _model(object_dataset)
    model.graph_rows = 16
    images, K, labels, T0, im_ids = _inputs(object_dataset, 1, 7)

    def timed(n=20):
        for _ in range(3):
            model(images=images, K=K, labels=labels, TCO=T0, n_iterations=5, im_ids=im_ids, materialize=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(images=images, K=K, labels=labels, TCO=T0, n_iterations=5, im_ids=im_ids, materialize=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    g = timed()
    model.graph_rows = 0
    e = timed()
    print(f"refiner call, 1 row x 5 iterations: hipGraph replay {g:.2f} ms, eager launches {e:.2f} ms")
    assert g < 1.25 * e
