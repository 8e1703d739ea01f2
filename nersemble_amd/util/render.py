"""Trajectory rendering -- mirror of the reference's ``util/render.py:13-72`` (``render_trajectory_video(model, cameras,
output_path, rendered_resolution_scaling_factor, render_channels, seconds)``): every camera of the trajectory is
rendered with ``model.get_outputs_for_camera_ray_bundle`` and each requested channel goes to its own video, whose path
is ``output_path.format(r=channel)``.

1-channel outputs are replicated to grey RGB; ``depth`` is turbo colour-mapped (inverted) between 0.8 and 1.2 m x the
dataset's scale factor 9 and faded by the accumulation, as the reference hard-codes (:47-54).

Container: the reference writes mp4 through ``mediapy`` (needs ffmpeg); neither is installed here, so when mediapy
cannot be imported the frames are written as numbered PNGs into ``<output_path without extension>/`` instead, and the
function returns the list of paths it wrote."""
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from .colormaps import ColormapOptions, apply_depth_colormap


class _PngSequenceWriter:
    def __init__(self, path: str, shape, fps: float):
        self.folder = os.path.splitext(path)[0]
        os.makedirs(self.folder, exist_ok=True)
        self.shape, self.fps, self.count = shape, fps, 0
        with open(os.path.join(self.folder, "fps.txt"), "w") as f:
            f.write(f"{fps}\n")

    def add_image(self, image: np.ndarray) -> None:
        from PIL import Image
        frame = (np.clip(image, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8) if image.dtype != np.uint8 else image
        Image.fromarray(frame).save(os.path.join(self.folder, f"frame_{self.count:05d}.png"))
        self.count += 1

    def close(self) -> str:
        return self.folder


class _MediapyWriter:
    def __init__(self, path: str, shape, fps: float):
        import mediapy
        self.path = path
        self.writer = mediapy.VideoWriter(path=path, shape=shape, fps=fps)
        self.writer.__enter__()

    def add_image(self, image: np.ndarray) -> None:
        self.writer.add_image(image)

    def close(self) -> str:
        self.writer.__exit__()
        return self.path


def _open_writer(path: str, shape, fps: float):
    try:
        return _MediapyWriter(path, shape, fps)
    except ImportError:
        return _PngSequenceWriter(path, shape, fps)


def render_trajectory_video(model, cameras, output_path: str, rendered_resolution_scaling_factor: float = 1.0,
                            render_channels: Optional[List[str]] = None, seconds: Optional[float] = None) -> List[str]:
    render_channels = ["rgb"] if render_channels is None else render_channels
    fps = 24 if seconds is None else len(cameras) / seconds
    cameras.rescale_output_resolution(rendered_resolution_scaling_factor)
    cameras = cameras.to(model.device)
    folder = os.path.dirname(output_path)
    if folder:
        os.makedirs(folder, exist_ok=True)

    writers: Dict[str, object] = {}
    for camera_idx in range(cameras.size):
        bundle = cameras.generate_rays(camera_indices=camera_idx)
        with torch.no_grad():
            outputs = model.get_outputs_for_camera_ray_bundle(bundle)
        for channel in render_channels:
            if channel not in outputs:
                raise KeyError(f"Could not find {channel} in the model outputs")
            if channel == "depth":
                frame = apply_depth_colormap(outputs["depth"], accumulation=outputs["accumulation"],
                                             near_plane=0.8 * 9, far_plane=1.2 * 9,
                                             colormap_options=ColormapOptions(colormap="turbo", invert=True))
            else:
                frame = outputs[channel]
                if frame.shape[-1] == 1:
                    frame = frame.expand(*frame.shape[:-1], 3)
            frame = frame.float().cpu().numpy()
            if channel not in writers:
                writers[channel] = _open_writer(output_path.format(r=channel), (frame.shape[0], frame.shape[1]), fps)
            writers[channel].add_image(frame)
    return [w.close() for w in writers.values()]
