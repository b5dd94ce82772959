"""BASELINE.json full-size checks (configs[2], 1024x1024x256) through size-independent properties:
the oracle cannot finish this size in seconds, so the post-network stages are checked against the
ANALYTIC answer of the lattice blob field (bench.py:device_blob_field): every blob whose skeleton
core lies inside the written frame becomes exactly one instance, every gated voxel of such a blob
carries that instance's id, ids are 1..K in first-appearance order, renumber is idempotent."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPE = (1024, 1024, 256)


def test_full_size_blob_field_known_answer():
    import bench
    from skoots_amd.lib import eval as E
    from skoots_amd import _ffi
    X, Y, Z = SHAPE
    field, ncell = bench.device_blob_field(SHAPE, (0, Z), DEV)
    assert ncell == 16 * 16 * 16

    def inject(_, origin, eff):
        x, y, z = origin
        return field[:, x:x + eff[0], y:y + eff[1], z:z + eff[2]]

    # the REAL network runs on every tile of a random image (the bench's launch geometry: 300x300x20 tiles,
    # batches of up to 64 as the free device memory allows, all 625 distinct origins); the blob field then replaces each tile's output, exactly as
    # bench.py does, so the analytic answer below must be unchanged by the network having run
    from skoots_amd import unet
    g = torch.Generator(device=DEV).manual_seed(1234)
    image = torch.randint(0, 256, SHAPE, generator=g, device=DEV, dtype=torch.uint8).to(torch.float16)
    seen = []

    def inject(out5, origin, eff):  # noqa: F811
        assert out5 is not None and tuple(out5.shape) == (5,) + tuple(eff)
        if len(seen) < 4 or origin[2] + eff[2] == Z:
            seen.append(out5)
        x, y, z = origin
        return field[:, x:x + eff[0], y:y + eff[1], z:z + eff[2]]

    res = E.eval_volume(image, unet.smoke_model(DEV), bench.SCALE, mean=127.5, std=73.9, inject=inject)
    assert len(seen) >= 4
    inst = res["instance_mask"]
    k = res["n_instances"]
    # analytic: 14 x 14 x 16 lattice cells keep their skeleton core inside the frame [50:974, 50:974, 5:251)
    assert k == 14 * 14 * 16
    assert int(inst.max()) == k
    # ids are 1..K in order of first appearance (C order)
    flat = inst.reshape(-1)
    nz = torch.nonzero(flat).flatten()
    vals = flat[nz]
    first = torch.full((k + 1,), flat.numel(), dtype=torch.int64, device=DEV)
    first.scatter_reduce_(0, vals.long(), nz, reduce="amin")
    assert bool((first[1:-1] < first[2:]).all())
    # every gated voxel (prob channel 0.95) of a cell maps to ONE id, and ids do not repeat across cells
    gated = field[4] > 0.9
    frame = torch.zeros(SHAPE, dtype=torch.bool, device=DEV)
    frame[50:X - 50, 50:Y - 50, 5:Z - 5] = True
    cell = ((torch.arange(X, device=DEV) // 64).view(-1, 1, 1) * 16 + (torch.arange(Y, device=DEV) // 64).view(1, -1, 1)) * 16 \
        + (torch.arange(Z, device=DEV) // 16).view(1, 1, -1)
    sel = gated & frame & (inst > 0)
    c, i = cell[sel].long(), inst[sel].long()
    lo = torch.full((ncell,), 2 ** 31, dtype=torch.int64, device=DEV).scatter_reduce_(0, c, i, reduce="amin")
    hi = torch.zeros((ncell,), dtype=torch.int64, device=DEV).scatter_reduce_(0, c, i, reduce="amax")
    live = hi > 0
    assert int(live.sum()) == k and bool((lo[live] == hi[live]).all())
    assert torch.unique(hi[live]).numel() == k
    # gated voxels of the surviving cells inside the frame are all assigned
    unassigned = gated & frame & (inst == 0)
    assert not bool(live[cell[unassigned].long()].any())
    # renumber is idempotent
    st = res["state"]
    before = st.instance.clone()
    st.renumber()
    assert torch.equal(st.instance, before)


def test_full_size_dense_field_follow_matches_oracle_on_reference_crops():
    """The worst case of the follow kernel at configs[2]'s size: bench.device_blob_field(dense=True) gates EVERY voxel
    in and gives it a non-zero vector, so all 268 M voxels make nine dependent hops (the default field is ~94 %
    background, whose voxels leave after one read).  The oracle cannot do 63 crops of 500x500x50 in seconds, but it can
    do two: for two crops of the reference's own grid (eval.py:248-258; one interior, one clamped to the far corner) the
    kernel's output must equal oracle.vector_to_embedding(N=10) + index_skeleton_by_embed bit for bit on the voxels
    that crop writes last.  Prints the kernel's byte rates for DESIGN.md."""
    import time

    import bench
    from oracle import pipeline as O
    from skoots_amd.lib import cropper
    from skoots_amd.lib.eval import ASSIGN_CROP, ASSIGN_OVERLAP
    from skoots_amd.parallel import ShardedVolume
    X, Y, Z = SHAPE
    field, _ = bench.device_blob_field(SHAPE, (0, Z), DEV, dense=True)
    assert float((field[4] > 0.8).float().mean()) == 1.0 and bool((field[0:3].abs().amax(dim=0) > 0).all())

    def inject(_, origin, eff):
        x, y, z = origin
        return field[:, x:x + eff[0], y:y + eff[1], z:z + eff[2]]

    sv = ShardedVolume(SHAPE, 0, 1, DEV)
    res = sv.run(torch.zeros(SHAPE, dtype=torch.float16, device=DEV), None, bench.SCALE, 0.0, 1.0, inject=inject)
    state, labels = res["state"], res["labels"]
    del field
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inst = state.assign(bench.SCALE, labels=labels)          # un-renumbered ids = label values
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"\nfollow+assign, dense field, {X}x{Y}x{Z}: {dt * 1e3:.2f} ms incl. launch = "
          f"{64.0 * X * Y * Z / dt / 1e9:.0f} GB/s at SURVEY's 64 B/voxel (every voxel makes 9 hops here)")
    frame = inst[50:X - 50, 50:Y - 50, 5:Z - 5]
    assert float((frame > 0).float().mean()) > 0.05             # the swirl does carry voxels onto skeletons
    vec = state.vectors_planar().cpu()                            # (3, X, Y, Z) fp16: the reference's zarr layout
    lab_cpu = labels.cpu()
    eff = cropper.clamp_crop_(list(ASSIGN_CROP), SHAPE)
    own = [cropper.owner_table(dm, c, o) for dm, c, o in zip(SHAPE, eff, ASSIGN_OVERLAP)]
    origins = cropper.crop_origins(SHAPE, list(ASSIGN_CROP), ASSIGN_OVERLAP)
    picks = [o for o in origins if o == (400, 400, 80)] + [origins[-1]]
    assert len(picks) == 2 and picks[1] == (X - eff[0], Y - eff[1], Z - eff[2])
    scale = torch.tensor(bench.SCALE)
    for (ox, oy, oz) in picks:
        v = vec[:, ox:ox + eff[0], oy:oy + eff[1], oz:oz + eff[2]].unsqueeze(0)
        emb = O.vector_to_embedding(scale, v, N=O.FOLLOW_N)                       # eval.py:271-273
        emb = emb + torch.tensor([ox, oy, oz], dtype=emb.dtype).view(1, 3, 1, 1, 1)  # eval.py:274-276
        want = O.index_skeleton_by_embed(lab_cpu.unsqueeze(0).unsqueeze(0), emb)[0, 0]
        sel = [torch.from_numpy(np.nonzero(t == o)[0]) for t, o in zip(own, (ox, oy, oz))]   # written last by this crop
        assert all(s.numel() > 0 for s in sel)
        gx, gy, gz = sel
        got = inst[gx[0]:gx[-1] + 1, gy[0]:gy[-1] + 1, gz[0]:gz[-1] + 1].cpu()
        w = want[gx[0] - ox:gx[-1] + 1 - ox, gy[0] - oy:gy[-1] + 1 - oy, gz[0] - oz:gz[-1] + 1 - oz]
        assert torch.equal(got, w.to(got.dtype)), (ox, oy, oz)
