"""CPU restatement of pinhole ray generation as the reference's datamanager obtains it -- TEST INFRASTRUCTURE ONLY
(the checker of csrc/raygen.hip and of nersemble_amd/cameras.py).

Call sites in the reference: ``NeRSembleVanillaDataManager.next_train`` -> ``self.train_ray_generator(ray_indices)``
(datamanager/nersemble_datamanager.py:76-81); cameras are ``CameraType.PERSPECTIVE`` with every distortion parameter zero
(dataparser/nersemble_dataparser.py:237-244).  The arithmetic itself lives in nerfstudio 0.3.1
(``nerfstudio/cameras/cameras.py::Cameras._generate_rays_from_coords``, ``model_components/ray_generators.py``), which is
not vendored and not installable here: PARITY UNPINNED -- restated from its published behaviour:

  * ``RayGenerator``: ``coords = image_coords[y, x]`` = pixel index + 0.5 (``get_image_coords(pixel_offset=0.5)``);
  * camera-frame direction ``((x - cx) / fx, -(y - cy) / fy, -1)`` (OpenGL: x right, y up, looking down -z);
  * the same for the pixels one step right and one step down; all three rotated by ``c2w[:3, :3]`` and normalised;
  * ``origins = c2w[:3, 3]``; ``pixel_area = |d - d_right| * |d - d_down|``.

float32 with the operation order of csrc/raygen.hip (numpy does not contract)."""
import numpy as np

f32 = np.float32


def _unit_dirs(rot, fx, fy, cx, cy, y, x):
    d0 = (x - cx) / fx
    d1 = -((y - cy) / fy)
    d2 = np.full_like(d0, -1.0)
    w = [(rot[:, i, 0] * d0 + rot[:, i, 1] * d1) + rot[:, i, 2] * d2 for i in range(3)]
    n = np.sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2])
    return np.stack([w[0] / n, w[1] / n, w[2] / n], axis=-1).astype(f32)


def generate_rays(camera_to_worlds, fx, fy, cx, cy, camera_indices, ys, xs):
    """camera_to_worlds [N,3,4]; fx, fy, cx, cy [N]; camera_indices [R] int; ys, xs [R] pixel coordinates WITH the +0.5
    pixel-centre offset.  Returns (origins [R,3], directions [R,3], pixel_area [R,1]) float32."""
    c2w = np.asarray(camera_to_worlds, f32).reshape(-1, 3, 4)
    c = np.asarray(camera_indices).reshape(-1).astype(np.int64)
    rot = c2w[c, :, :3]
    fx, fy, cx, cy = (np.asarray(t, f32).reshape(-1)[c] for t in (fx, fy, cx, cy))
    ys, xs = np.asarray(ys, f32).reshape(-1), np.asarray(xs, f32).reshape(-1)
    d = _unit_dirs(rot, fx, fy, cx, cy, ys, xs)
    dx = _unit_dirs(rot, fx, fy, cx, cy, ys, xs + f32(1))
    dy = _unit_dirs(rot, fx, fy, cx, cy, ys + f32(1), xs)

    def norm(a):
        return np.sqrt((a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]) + a[:, 2] * a[:, 2])

    area = (norm(d - dx) * norm(d - dy)).astype(f32)[:, None]
    return c2w[c, :, 3].copy(), d, area


def ray_generator(camera_to_worlds, fx, fy, cx, cy, ray_indices):
    """nerfstudio ``RayGenerator.forward``: (camera, y, x) integer triples -> rays through the pixel centres."""
    idx = np.asarray(ray_indices).astype(np.int64)
    return generate_rays(camera_to_worlds, fx, fy, cx, cy, idx[:, 0], idx[:, 1].astype(f32) + f32(0.5),
                         idx[:, 2].astype(f32) + f32(0.5))
