"""Colour back-projection and the thinning offsets of the export on the device.

    color_projection      instant_nsr/utils/coloring_utils.py:91-138   (save_mesh, mesh_utils.py:53-55)
    get_offset_mask       instant_nsr/utils/thinning_utils.py:96-193   (thinning_processing, :236)

The reference loops over vertices / skeleton pixels in Python and fires one `mesh_raycast.raycast`
per item; every ray is parallel to z.  Here the rays of a pass are one launch over an xy grid of
the triangles (ops.zray_cast), the silhouette (pytorch3d MaskRenderer) one rasterising launch, the
19x19 elliptic erosion one launch, the k = 8 nearest-neighbour fill one launch.  The order-dependent
part of get_offset_mask ("first ray to reach a vertex decides its offset") is kept exactly: every
candidate carries its position in the reference's loop order and the smallest valid one wins.

The host-side image glue (PIL LANCZOS resizes of colour / mask PNGs) stays with the caller: the
functions take the 2048^2 uint8 images the reference would have loaded.
"""
import numpy as np
import torch

from .. import ops


def _tris(verts, faces):
    """np.array(vertices[faces], dtype='f4') (coloring_utils.py:92-93)."""
    return verts.to(torch.float32)[faces.long()].contiguous()


def render_mask(verts, faces, res=2048, scale=2.0):
    """MaskRenderer.render(vertices * 2, faces) (coloring_utils.py:22-41, :99)."""
    return ops.raster_mask(_tris(verts, faces), scale, res)


def get_color_from_image(pc, color_map, back=False):
    """coloring_utils.py:67-86: nearest pixel of the (res,res,C) map under (x, -y) (x mirrored for
    the back view); np.round = round-half-to-even = torch.round."""
    res = color_map.shape[0]
    xy = pc[:, 0:2].clone().to(torch.float64)
    if back:
        xy[:, 0] *= -1
    xy[:, 1] *= -1
    xy = (xy + 0.5) * (res - 1)
    xi = torch.round(xy[:, 0]).long().clamp(0, color_map.shape[1] - 1)
    yi = torch.round(xy[:, 1]).long().clamp(0, color_map.shape[0] - 1)
    return color_map[yi, xi]


def load_color(color_u8, mask_u8, ksize=19):
    """coloring_utils.py:61-65 without the file reads: erode the mask with the 19x19 ellipse, stack
    as alpha, float32 / 255."""
    mask = ops.erode_ellipse_u8(mask_u8.contiguous(), ksize)
    rgba = torch.cat([color_u8, mask[:, :, None]], 2)
    return rgba.to(torch.float32) / 255.0


@torch.no_grad()
def color_projection(verts, faces, color_front_u8, mask_front_u8, color_back_u8, res=2048):
    """coloring_utils.py:91-138.  verts (V,3) float64 in the save_mesh frame (x right, y up, z
    front, inside [-0.5, 0.5]), faces (F,3); color_*_u8 (res,res,3) and mask_front_u8 (res,res): the
    LANCZOS-resized PNGs of <uid>/mv/{color,mask}.  Returns (V,3) float64 colours."""
    dev = verts.device
    V = verts.shape[0]
    faces_i = faces.to(torch.int32).contiguous()
    tris = _tris(verts, faces)
    lo = tris[..., :2].reshape(-1, 2).amin(0).tolist()
    hi = tris[..., :2].reshape(-1, 2).amax(0).tolist()
    grid = ops.ZGrid(tris, lo, hi)
    v32 = verts.to(torch.float32).contiguous()              # mesh_raycast parses the source as float
    all_ids = torch.arange(V, dtype=torch.int32, device=dev)
    vert_colors = torch.zeros(V, 4, dtype=torch.float64, device=dev)

    mask_front = torch.minimum(mask_front_u8, render_mask(verts, faces, res, 2.0))
    # front: vertices whose pixel is inside the eroded mask and that nothing covers from +z
    col = get_color_from_image(verts, load_color(color_front_u8, mask_front))
    cand = torch.nonzero(col[:, 3] > 0)[:, 0]
    cnt, _, _, t_far, _ = ops.zray_cast(grid, faces_i, v32[cand], +1, all_ids[cand].contiguous())
    ok = (cnt > 0) & (t_far == 0)
    vert_colors[cand[ok]] = col[cand[ok]].to(torch.float64)
    # back: the still uncoloured ones, mirrored mask / image, nothing behind them along -z
    mask_back = mask_front.flip(1)
    rest = torch.nonzero(vert_colors[:, 3] == 0)[:, 0]
    col = get_color_from_image(verts[rest], load_color(color_back_u8, mask_back.contiguous()), True)
    keep = col[:, 3] > 0
    cand, col = rest[keep], col[keep]
    cnt, _, _, t_far, _ = ops.zray_cast(grid, faces_i, v32[cand], -1, all_ids[cand].contiguous())
    ok = (cnt > 0) & (t_far == 0)
    vert_colors[cand[ok]] = col[ok].to(torch.float64)
    # the remaining vertices: 8 nearest coloured vertices in the xy plane
    unknown = vert_colors[:, 3] == 0
    known = ~unknown
    if bool(unknown.any()) and bool(known.any()):
        rgb = ops.knn8_blend(verts[unknown][:, :2], verts[known][:, :2],
                             vert_colors[known][:, :3].to(torch.float32))
        vert_colors[unknown, 0:3] = rgb.to(torch.float64)
    return vert_colors[:, 0:3]


@torch.no_grad()
def get_offset_mask(verts, faces, thin_coords, coord_dists, min_thickness, type="double"):
    """thinning_utils.py:96-193.  thin_coords (n,2) skeleton pixel positions in mesh xy units,
    coord_dists (n,) half-thickness read from the distance map.  Returns (offset_values (V,3)
    float64, offset_mask (V,) bool) — the Dirichlet data of the harmonic deformation."""
    if type not in ("double", "front", "back"):
        raise ValueError(type)
    dev = verts.device
    V, n = verts.shape[0], thin_coords.shape[0]
    faces_i = faces.to(torch.int32).contiguous()
    tris = _tris(verts, faces)
    lo = tris[..., :2].reshape(-1, 2).amin(0).tolist()
    hi = tris[..., :2].reshape(-1, 2).amax(0).tolist()
    grid = ops.ZGrid(tris, lo, hi)
    src = torch.cat([thin_coords.to(torch.float32), torch.ones(n, 1, device=dev)], 1)
    cnt0, _, f_near, _, f_far = ops.zray_cast(grid, faces_i, src, -1)
    hit = cnt0 > 0
    target = torch.maximum(torch.as_tensor(min_thickness, dtype=torch.float64, device=dev),
                           coord_dists.to(torch.float64) * 2)
    idx = torch.arange(n, device=dev)
    keys, verts_id, valid, values = [], [], [], []

    def probe(face_of_ray, sign, slot, scale, fix_only):
        """the three vertices of each ray's face: secondary ray from the vertex along `sign`."""
        f = face_of_ray.clamp(min=0).long()
        for j in range(3):
            vid = faces[f, j].long()
            key = idx * 6 + slot + j
            if fix_only:
                ok = hit
                val = torch.zeros(n, dtype=torch.float64, device=dev)
            else:
                coord = tris[f, j]                                     # f4 vertex of the triangle
                cnt, _, _, t_far, _ = ops.zray_cast(grid, faces_i, coord.contiguous(), sign,
                                                    vid.to(torch.int32).contiguous())
                # dist = coord.z - hit.z (front) / hit.z - coord.z (back), evaluated as the
                # reference does from the float32 hit point
                zh = coord[:, 2] + float(sign) * t_far
                dist = ((coord[:, 2] - zh) if sign < 0 else (zh - coord[:, 2])).to(torch.float64)
                ok = hit & (cnt > 0) & (dist > target) & (dist < 0.06)
                val = (dist - target) * scale * (-1.0 if sign < 0 else 1.0)
            keys.append(key); verts_id.append(vid); valid.append(ok); values.append(val)

    if type == "double":
        probe(f_near, -1, 0, 0.5, False)
        probe(f_far, +1, 3, 0.5, False)
    elif type == "front":
        probe(f_near, -1, 0, 1.0, False)
        probe(f_far, +1, 3, 0.0, True)
    else:
        probe(f_near, -1, 0, 0.0, True)
        probe(f_far, +1, 3, 1.0, False)
    keys, verts_id = torch.cat(keys), torch.cat(verts_id)
    valid, values = torch.cat(valid), torch.cat(values)
    big = torch.iinfo(torch.int64).max
    first = torch.full((V,), big, dtype=torch.int64, device=dev)
    first.scatter_reduce_(0, verts_id[valid], keys[valid], reduce="amin")
    win = valid & (keys == first[verts_id])
    offset_values = torch.zeros(V, 3, dtype=torch.float64, device=dev)
    offset_values[verts_id[win], 2] = values[win]
    return offset_values, first != big
