#!/usr/bin/env python3
"""The reference's run.py scene (reference: run.py:20-41) on the MI355X path.

    python examples/run.py /path/to/objects/T-Rex.obj output.png [size] [--device-model]

``--device-model``: the transforms of the scene — two rotations with their vertex-normal
recomputation (1.2 s of numpy loops on the host for T-Rex) and the fit — run on a
``DeviceModel`` kept in HBM (rotate / normals to 1e-5 of the host Model, the rest bit for bit).

Model -> AdvancedPixelBufferFiller (HIP, GuroIllumination fused into the raster kernel's stores)
-> flip + uint8 (HIP) -> PNG.
The .obj / .mtl / texture assets are the reference's; they are not part of this repository.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cython3dmodelrenderer_amd import Renderer                                   # noqa: E402
from cython3dmodelrenderer_amd.data_structures import DeviceModel, Model          # noqa: E402
from cython3dmodelrenderer_amd.illumination import GuroIllumination               # noqa: E402
from cython3dmodelrenderer_amd.pixel_buffer_filler import AdvancedPixelBufferFiller  # noqa: E402
from cython3dmodelrenderer_amd.scenes import fit_model                            # noqa: E402


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    obj, out = args[0], args[1]
    size = int(args[2]) if len(args) > 2 else 1024
    t0 = time.perf_counter()
    model = Model.read_model(obj, recalculate_normals="--device-model" not in sys.argv)
    if model._colors_by_triangles is None:
        model.set_uniform_color()            # untextured models: white, as the py renderer does
    if "--device-model" in sys.argv:
        model = DeviceModel(model)
    model.rotate([-90, 180, 0])
    model.rotate([10, -80, 0])
    fit_model(model)
    t1 = time.perf_counter()
    filler = AdvancedPixelBufferFiller(size, size, fov=45, n_threads=8)
    renderer = Renderer(filler, GuroIllumination([0, 0, 1]), None, *filler.get_size(), on_device="fused")
    renderer.render(model)
    image_bgr = filler.present_u8().cpu().numpy()          # == image[::-1].astype('uint8')
    t2 = time.perf_counter()
    from PIL import Image
    Image.fromarray(image_bgr[:, :, ::-1].copy(), "RGB").save(out)   # cv2.imwrite takes BGR; PIL wants RGB
    print(f"model {t1 - t0:.2f} s, render + shade + present {1e3 * (t2 - t1):.1f} ms "
          f"(first call: includes plan creation and upload), wrote {out}")


if __name__ == "__main__":
    main()
