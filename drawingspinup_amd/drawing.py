"""One drawing through the hot path: 6-view diffusion -> NSR reconstruction -> 24-frame
stylisation (mv.py -> recon.py -> test_stage1.py -> test_stage2.py of the reference), with the
stages handing tensors over in memory instead of PNG/OBJ files.

Synthetic inputs: a 512x512 RGBA "drawing" (low-pass random colour inside an
ellipse), random-init weights of the reference architectures.  Blender/Mixamo rigging between
recon and stylisation is an external manual tool in the reference (README.md:183-186); the
stylisation frames are synthetic colour / position / edge maps of the stated shapes.
"""
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from .entry.data import fill_holes
from .mv.pipeline import build_random_pipeline
from .mv.preprocess import pil_resize_rgba_u8, pil_resize_u8
from .nsr.system import OrthoData, OrthoNeuSSystem, VIEWS, inv_rt, rt_opengl2opencv, ideal_w2c
from .style.generators import build_model

STYLE_ARGS = dict(use_bias=False, tanh=True, append_smoothers=True, resnet_blocks=7,
                  filters=[32, 64, 128, 128, 128, 64], input_channels=6)   # config_stage{1,2}.yaml


def synthetic_drawing(seed, size=512, device="cuda"):
    """(4,size,size) float RGBA in [0,1]: 8x8-block colour noise inside a filled ellipse."""
    g = torch.Generator().manual_seed(seed)
    blocks = torch.rand(3, size // 8, size // 8, generator=g)
    rgb = F.interpolate(blocks[None], size=(size, size), mode="nearest")[0]
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing="ij")
    alpha = ((xx / 0.55) ** 2 + (yy / 0.8) ** 2 <= 1.0).float()
    return torch.cat([rgb * alpha + (1 - alpha), alpha[None]], 0).to(device)


def synthetic_frames(seed, n_frames=24, size=512, device="cuda"):
    """DatasetFullImages tensors (training/data.py:23-47): (n,6,H,W) = RGB[-1,1] + mask + pos-XY."""
    g = torch.Generator().manual_seed(seed + 1000)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing="ij")
    tex = F.interpolate(torch.rand(1, 3, 16, 16, generator=g), size=(size, size), mode="bilinear",
                        align_corners=False)[0]
    frames = []
    for f in range(n_frames):
        cx = 0.3 * np.sin(2 * np.pi * f / n_frames)
        mask = (((xx - cx) / 0.5) ** 2 + (yy / 0.75) ** 2 <= 1.0).float()
        rgb = (tex * 2 - 1) * mask + (1 - mask)
        pos = torch.stack([xx, yy]) * mask + (1 - mask)
        frames.append(torch.cat([rgb, mask[None], pos], 0))
    return torch.stack(frames).to(device)


def synthetic_edges(frames):
    """The `edge/NNNN.png` maps of run_render.py:31-57,117-120 for synthetic frames: per-channel
    3x3 Sobel magnitude of the position map (background set to 2), max over channels, > 0.3 is an
    edge; stored inverted (255 = no edge, 0 = edge) as `cv2.imwrite(..., 255 - edge)` does.
    frames (n,6,H,W) with the mask at channel 3 and pos-XY in [-1,1] at 4:6 -> (n,H,W) uint8."""
    mask = frames[:, 3:4]
    pos = (frames[:, 4:6] + 1) / 2                                  # the PNG's [0,1] values
    pos = torch.where(mask < 1, torch.full_like(pos, 2.0), pos)
    kx = torch.tensor([[-1.0, 0, 1], [-2, 0, 2], [-1, 0, 1]], device=frames.device)
    k = torch.stack([kx, kx.t()])[:, None]                           # (2,1,3,3): d/dx, d/dy
    p = F.pad(pos.reshape(-1, 1, *pos.shape[-2:]), (1, 1, 1, 1), mode="reflect")   # cv2 BORDER_REFLECT_101
    g = F.conv2d(p, k)
    val = (g[:, 0] ** 2 + g[:, 1] ** 2).sqrt().reshape(pos.shape[0], 2, *pos.shape[-2:]).amax(1)
    return torch.where(val > 0.3, 0, 255).to(torch.uint8)


def to_image_space(x):
    """custom_transforms.py:8-9 on the device: clip -> uint8."""
    return ((x.clamp(-1, 1) + 1) / 2 * 255).to(torch.uint8)


def _u8_hwc(t):
    """tensor2pil (mv.py:46-48) on the device: (C,H,W) float in [0,1] -> (H,W,C) uint8 with
    mul 255, add 0.5, clamp — what the reference's PNG hand-offs hold."""
    return (t.float() * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous()


class DrawingPipeline:
    """Holds the shared read-only weights (diffusion UNet/VAE/CLIP) and runs drawings."""

    def __init__(self, device="cuda", seed=0, mv_steps=75, nsr_steps=3000, n_frames=24,
                 with_clip=True, export_resolution=512, with_mv=True, with_contour=True, mesh_post=True,
                 with_matting=None, isnet_weights=None):
        self.device = torch.device(device)
        self.mv_steps, self.nsr_steps, self.n_frames = mv_steps, nsr_steps, n_frames
        self.style_batch = 4                 # frames per generator call
        self.time_substages = False          # bench.py: split the NSR stage into fit / export (the
        #                                      synchronisations are the CURRENT STREAM's: several drawings
        #                                      may be in flight on one GPU, one stream + one pipeline each)
        self.substage_seconds = {}
        self.fit_stream = None               # a stream for the NSR optimisation alone (see reconstruct)
        self.fit_gate = None                 # a semaphore shared by the pipelines of one GPU: how many
        #                                      drawings may be inside the NSR optimisation at a time
        self.export_resolution = export_resolution
        self.mesh_post = mesh_post           # the reference's export switches (remesh, smooth, cbp, shear)
        self.mv = build_random_pipeline(self.device, seed, with_clip=with_clip) if with_mv else None
        torch.manual_seed(seed + 1)
        self.gen1 = build_model("GeneratorJ_RIC", STYLE_ARGS, self.device).eval()
        self.gen2 = build_model("GeneratorJ", STYLE_ARGS, self.device).eval()
        # side-view matting (mv.py:113-150): the IS-Net forward on the four predicted side views.
        # With a DIS checkpoint its mattes are the side masks; with random weights the network RUNS
        # (its cost belongs to a drawing) and the filled-silhouette stand-in supplies the masks.
        self.isnet, self.isnet_trained = None, isnet_weights is not None
        if with_mv if with_matting is None else with_matting:
            from .mv import matting
            self.isnet = matting.load_isnet(isnet_weights, self.device, seed=seed + 3)
        self.last_side_mattes = None
        self.contour = None
        if with_contour:
            from .contour.predict import load_generator
            torch.manual_seed(seed + 2)
            self.contour = load_generator(None, self.device)
            self._calibrate_random_contour()

    @torch.no_grad()
    def _calibrate_random_contour(self, fraction=0.03):
        """Random-init weights put the contour probability near 0.5 everywhere: the whole image
        would be "contour", the inpainting front empty and the stage's host tail a no-op.  Shift
        the output bias so that ~3 % of a synthetic drawing's pixels exceed predict.py's 0.2
        threshold (thin-line coverage), i.e. the TELEA tail does representative work."""
        last = [m for m in self.contour.modules() if isinstance(m, torch.nn.Conv2d)][-1]
        d = synthetic_drawing(4242, device=self.device)
        x = torch.cat([d[:3] * d[3:4] + (1 - d[3:4]), d[3:4]], 0)[None]
        p = self.contour(x)[0, 0].float().clamp(1e-6, 1 - 1e-6)
        logit = torch.log(p / (1 - p)).flatten()
        q = torch.quantile(logit[::5], 1.0 - fraction)
        last.bias += float(np.log(0.2 / 0.8)) - float(q)

    def shared_modules(self):
        mods = [self.gen1, self.gen2]
        if self.mv is not None:
            mods += [self.mv.unet, self.mv.vae]
            if self.mv.image_encoder is not None:
                mods.append(self.mv.image_encoder)
        if self.contour is not None:
            mods.append(self.contour)
        if self.isnet is not None:
            mods.append(self.isnet)
        return mods

    # ---------------------------------------------------------------- stage 1: predict.py
    @torch.no_grad()
    def remove_contour(self, drawing_rgba, threshold=0.2, radius=3):
        """1_lama_contour_remover/predict.py:46-66: the drawing composited on white + its alpha ->
        FFC-ResNet generator (device) -> contour mask (prob > 0.2), inpaint mask
        max(contour, 255 - alpha) -> TELEA inpainting of the uint8 image (host code in
        libdsu_hip.so, as the reference's cv2.inpaint: fast marching is serial) -> RGBA with the
        input alpha, handed to the diffusion stage in memory instead of `_inpainted.png`."""
        if self.contour is None:
            return drawing_rgba
        from .contour.predict import inpaint
        rgb, a = drawing_rgba[:3], drawing_rgba[3:4]
        x = torch.cat([rgb * a + (1 - a), a], 0)[None]
        prob = self.contour(x)[0, 0].float()
        contour = prob > threshold
        self.last_contour_masks = (contour, contour | (a[0] < 1.0))
        # predict.py:55-62 on the host (uint8 truncation as .astype('uint8') there)
        xh = (x[0] * 255).to(torch.uint8).permute(1, 2, 0).contiguous().cpu().numpy()
        img, alpha = np.ascontiguousarray(xh[:, :, :3]), xh[:, :, 3]
        mask = np.maximum(contour.to(torch.uint8).mul(255).cpu().numpy(), 255 - alpha)
        inpainted = inpaint(img, mask, radius)
        out = np.concatenate([inpainted, alpha[:, :, None]], 2)
        return torch.from_numpy(out).to(self.device).permute(2, 0, 1).float() / 255.0

    # ---------------------------------------------------------------- stage 2a: mv.py
    @torch.no_grad()
    def multiview(self, drawing_rgba, seed):
        """SingleImageDataset (white-background 256x256 x6 views) -> 12-sample batch ->
        pipeline (mv.py:70-86).  Returns normals (6,3,256,256), colours (6,3,256,256) in [0,1]."""
        # single_image_dataset.py:97-130 as entry/data.py load_image_rgba restates it: the RGBA image
        # through Pillow's default (bicubic, premultiplied-alpha) resize to 256^2 — the resampler of
        # mv/preprocess.py, bit for bit Pillow's (tests/test_mv_preprocess.py) — then composited on
        # white in f32
        small = pil_resize_rgba_u8(_u8_hwc(drawing_rgba), (256, 256), "bicubic").float() / 255.0
        img = (small[..., :3] * small[..., 3:4] + (1.0 - small[..., 3:4])).permute(2, 0, 1)[None]
        imgs_in = img.expand(12, -1, -1, -1).contiguous()
        g = torch.Generator(device=self.device).manual_seed(seed)
        out = self.mv(imgs_in, generator=g, guidance_scale=1.0, output_type="pt", eta=1.0,
                      num_inference_steps=self.mv_steps)
        return out[:6], out[6:]

    # ---------------------------------------------------------------- stage 2b: recon.py
    def reconstruct(self, normals, colors, drawing_rgba, seed):
        """OrthoDatasetBase from in-memory mv outputs (ortho.py:54-97: 1024^2 images, normals
        from the normal maps rotated to the front camera's world frame, masks), then the NSR
        optimisation and the export (2 x 512^3 SDF volumes -> smoothing -> marching cubes)."""
        dev = self.device
        size = 1024
        # mv.py:105-106: tensor2pil(view).resize((1024, 1024), Image.LANCZOS), saved as PNG and read
        # back as k / 255 (ortho.py:54-97) — the same 8-bit images, held in memory
        up8 = lambda t: torch.stack([pil_resize_u8(_u8_hwc(v), (size, size), "lanczos") for v in t])
        col8, nrm8 = up8(colors), up8(normals)                             # (6,1024,1024,3) uint8
        col = col8.float() / 255.0
        nrm = nrm8.float() / 255.0 * 2 - 1                                 # img2normal
        alpha = F.interpolate(drawing_rgba[3:4][None], size=(size, size), mode="nearest")[0, 0]
        # front: the drawing's alpha, back: mirrored (mv.py:113-116); side views: matte of the
        # predicted colour image (distance to the white background, entry/data.py
        # side_mask_from_prediction — the reference runs a CPU ONNX matting model there)
        mattes = None
        if self.time_substages:
            torch.cuda.current_stream(dev).synchronize()
        t_m = time.time()
        if self.isnet is not None:
            # remove_background (mv.py:134-150) on the 8-bit side-view images, in memory: (x / 255 -
            # 0.5) / 1.0 in CHW, the network, clip to [0, 1], * 255 -> uint8.  The four views go
            # through as one batch (the reference calls the session once per view; the network is
            # per-image in eval mode, and the convolutions fill the chip better at B = 4)
            with torch.no_grad():
                u8 = col8[[1, 2, 4, 5]].permute(0, 3, 1, 2)                # the LANCZOS 1024^2 images themselves
                mattes = (self.isnet(u8.float() / 255.0 - 0.5).clamp(0, 1) * 255).to(torch.uint8)[:, 0]
            self.last_side_mattes = mattes
        if self.time_substages:
            torch.cuda.current_stream(dev).synchronize()
            self.substage_seconds["nsr_matting"] = time.time() - t_m
        if mattes is not None and self.isnet_trained:
            side = torch.zeros(col.shape[:3], dtype=torch.bool, device=dev)
            side[[1, 2, 4, 5]] = mattes > 127                              # load_mask's threshold (ortho.py)
        else:
            side = (1.0 - col).amax(-1) > 12.0 / 255.0
            side = torch.stack([fill_holes(m) for m in side])              # a filled silhouette, as a matte is
        masks = torch.stack([alpha > 0.5, side[1], side[2], alpha.flip(1) > 0.5, side[4], side[5]])
        nrm = nrm * masks[..., None]
        front = torch.from_numpy(inv_rt(rt_opengl2opencv(ideal_w2c("front")))[:3, :3]).float().to(dev)
        n_cv = nrm * torch.tensor([1.0, -1.0, -1.0], device=dev)           # normal_opengl2opencv
        n_world = n_cv @ front.T
        poses = torch.stack([torch.from_numpy(inv_rt(rt_opengl2opencv(ideal_w2c(v)))).float()
                             for v in VIEWS])
        data = OrthoData(col, masks, n_world, poses, dev)
        system = OrthoNeuSSystem(device=dev, seed=seed)
        if self.fit_gate is not None:
            self.fit_gate.acquire()
        try:
            t0 = time.time()
            if self.fit_stream is not None:
                # the optimisation on a stream of its own (bench.py --fit-priority: a lower priority
                # than the stream of the latency-bound stages of the other drawings in flight)
                outer = torch.cuda.current_stream(dev)
                self.fit_stream.wait_stream(outer)
                with torch.cuda.stream(self.fit_stream):
                    system.fit(data, max_steps=self.nsr_steps)
                outer.wait_stream(self.fit_stream)
                self._fit_on_own_stream = True
            else:
                system.fit(data, max_steps=self.nsr_steps)
            if self.time_substages or self.fit_gate is not None:
                torch.cuda.current_stream(dev).synchronize()
        finally:
            if self.fit_gate is not None:
                self.fit_gate.release()
        t1 = time.time()
        # export (neus_ortho.py:183-200): smoothed binary volumes, front-mask cutting with the
        # drawing's own alpha (char/mask.png, rotated as ortho.py:155-156), marching cubes, colours
        front = (F.interpolate(drawing_rgba[3:4][None], size=(size, size), mode="nearest")[0, 0] * 255
                 + 0.5).to(torch.uint8)                                 # mask_front.resize(res, NEAREST)
        # the switches of configs/neuralangelo-ortho-wmask.yaml:12-20,45-46 as recon.py runs a uid
        # that is not in the thinning list: remeshing to 50 000 faces inside the fine stage
        # (geometry.py:63-64), no texture-network colours when colour back-projection is on
        # (neus.py:224-236)
        mesh = system.export_mesh(torch.rot90(front, k=-1, dims=(0, 1)).contiguous(), self.export_resolution,
                                  with_colors=not self.mesh_post, face_count=50000 if self.mesh_post else None)
        self.last_mesh = mesh
        if self.time_substages:
            torch.cuda.current_stream(dev).synchronize()
        t2 = time.time()
        if self.mesh_post and mesh["faces"].shape[0]:
            # save_mesh (mesh_utils.py:25-73): Laplacian smoothing, colour back-projection from the
            # predicted front / back views (2048^2, coloring_utils.py:62,100), shear, ortho scale;
            # the OBJ text write itself stays outside (file I/O)
            from .nsr.mesh import post_process_mesh
            # coloring_utils.py:62,100: the 1024^2 PNGs (colour views; the NEAREST-resized front mask)
            # through Image.resize((2048, 2048), LANCZOS)
            big = lambda u8: pil_resize_u8(u8, (2048, 2048), "lanczos")
            cbp = {"color_front": big(col8[0]), "color_back": big(col8[3]),
                   "mask_front": big(front[:, :, None])[:, :, 0].contiguous()}
            v, f, c = post_process_mesh(mesh["verts"], mesh["faces"], None, ortho_scale=1.35,
                                        smoothing=True, shearing=True, color_back_projection=cbp)
            self.last_mesh_post = {"verts": v, "faces": f, "colors": c}
        if self.time_substages:
            torch.cuda.current_stream(dev).synchronize()
            self.substage_seconds.update({"nsr_fit": t1 - t0, "nsr_export": t2 - t1,
                                          "nsr_post": time.time() - t2})
        if getattr(self, "_fit_on_own_stream", False):
            # tensors of the fit's stream were read by this stream's export: they go back to the fit
            # stream's pool only once this stream is done with them
            torch.cuda.current_stream(dev).synchronize()
        return system, mesh["binary"]

    # ---------------------------------------------------------------- stage 3: test_stage1/2.py
    @torch.no_grad()
    def stylize(self, frames, edges=None):
        """frames (n,6,H,W); edges (n,H,W) uint8 (255 = no edge) or None.  Stage 1, quantised to
        uint8 like the PNG hand-off, the edge map painted black onto it (DatasetFullImages with
        use_edge: overlap_edge_on_img, data.py:34-37, custom_transforms.py:31-36), then stage 2 on
        the re-normalised RGB (+ mask + pos) — test_stage1.py / test_stage2.py loop over the frames
        one by one; the generators are per-image functions in eval mode, so the frames go through
        in chunks of `style_batch` (same per-frame results, and the 64^2 / 128^2 levels of the
        U-net get enough tiles to fill the chip)."""
        outs = []
        for i in range(0, frames.shape[0], self.style_batch):
            x = frames[i:i + self.style_batch]
            s1 = self.gen1(x)
            q8 = to_image_space(s1)                                         # PNG round trip
            if edges is not None:
                e = edges[i:i + self.style_batch]
                q8 = torch.where((e < 255)[:, None], torch.zeros_like(q8), q8)
            q = q8.float() / 255.0 * 2 - 1
            s2 = self.gen2(torch.cat([q, x[:, 3:]], 1))
            outs.append(torch.cat([to_image_space(s2), (x[:, 3:4] * 255).to(torch.uint8)], 1))
        return torch.cat(outs)

    def run(self, seed):
        drawing = synthetic_drawing(seed, device=self.device)
        normals, colors = self.multiview(drawing, 123456 + seed)
        system, inside = self.reconstruct(normals, colors, drawing, 123456 + seed)
        fr = synthetic_frames(seed, self.n_frames, device=self.device)
        frames = self.stylize(fr, synthetic_edges(fr))
        return {"views": colors, "inside_voxels": inside.sum(), "frames": frames}
