"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's render -> coarse -> refine -> score hot path
(SURVEY.md section 8).  Nothing under this package is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / the timed CPU baseline.  The product
(``megapose6d_amd``) never imports ``oracle`` and fails loudly when its HIP
library is missing.

Parity status (also stated in DESIGN.md):
* geometry / pose math, crop boxes, K_crop, SO(3) grid, pose update, orchestration:
  PINNED against the reference's own Python imported from /root/reference
  (``oracle/ref_import.py`` + ``oracle/make_golden.py`` -> ``tests/golden/*.npz``).
* CNN backbones: PINNED against the reference's own module classes
  (``models/torchvision_resnet.py``, ``models/wide_resnet.py``) with seeded weights.
* roi_align (torchvision 0.12.0), unitquat_to_rotmat (roma), SE3 (pinocchio),
  PLY loading (trimesh), look-at cameras (panda3d NodePath): third-party code that
  is NOT under /root/reference; restated from their published algorithms --
  "parity unpinned" for those five.
* rasteriser: the reference delegates to Panda3D/OpenGL which is absent; the C
  software rasteriser in ``oracle/raster.c`` DEFINES the pixel contract --
  "parity unpinned vs Panda3D pixels".
"""
