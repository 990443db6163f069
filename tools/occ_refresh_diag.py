"""Why `visited occupied cell  =>  value changed` is not a usable property of the occupancy
refresh (the red GPU test of round 4): dump, for a few seeds, the occupied cells that the
regular refresh provably visited (the call exports its cells) and whose value nevertheless stayed
bit-identical, with the old value, the alpha the refresh computed and the selection sizes.
usage (GPU box): python tools/occ_refresh_diag.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from drawingspinup_amd.nsr.system import OrthoData, OrthoNeuSSystem
dev = torch.device("cuda:0")
for seed in (12, 13, 14, 15):
    ds = OrthoData.synthetic_sphere(256, device=dev)
    sysm = OrthoNeuSSystem(device=dev, seed=seed)
    sysm.dataset = ds
    sysm.step_mode = "native"
    for _ in range(2):
        sysm.training_step()
    drv, grid = sysm._native, sysm.model.occupancy_grid
    assert drv.occ_refresh(grid, 641, True, 0.01, 0.95)                 # warm-up form first, as the test did
    occs0, was_on = grid.occs.clone(), grid.binary_u8().bool().clone()
    ex = {}
    assert drv.occ_refresh(grid, 640, False, 0.01, 0.95, export=ex)
    cells = ex["cells"].long()
    n = grid.num_cells
    visited = torch.zeros(n, dtype=torch.bool, device=dev)
    visited[cells[cells >= 0]] = True
    occ_slots = cells[n // 4:]
    sel = occ_slots[occ_slots >= 0]
    unchanged = was_on & visited & (grid.occs == occs0)
    print(f"seed {seed}: occupied {int(was_on.sum())}  selected {sel.numel()}  "
          f"selection == nonzero(binary): {bool(torch.equal(sel, torch.nonzero(was_on)[:, 0]))}  "
          f"occupied and not visited: {int((was_on & ~visited).sum())}  "
          f"occupied, visited, value bit-identical: {int(unchanged.sum())}")
    for c in torch.nonzero(unchanged)[:5, 0].tolist():
        print(f"    cell {c}: occs0 = {float(occs0[c]):.9g}  (0.95 occs0 = {float(occs0[c]) * 0.95:.9g}; "
              f"kept only if the new alpha == occs0 bit for bit)")
