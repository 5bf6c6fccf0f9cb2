"""CPU oracle for the MI355X cataract-segmentation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / CPU baseline.  The product
package (``miccai2021_cataract_semantic_segmentation_amd``) never imports it
and raises if its HIP library is missing.

What it is: a plain-PyTorch fp32 CPU restatement of the reference's algorithm
for the hot path named by BASELINE.json (segmentation-network forward/backward,
Lovasz-Softmax, CE, metrics, Adam step).  Every function cites the reference
file:line (relative to RViMLab/MICCAI2021_Cataract_semantic_segmentation) it
follows.

Pinning status
--------------
* PINNED against fixtures in ``tests/golden/*.npz`` produced by importing the
  real reference in the build container (``tests/golden/make_golden.py``):
  OCR head modules, DeepLabv3+ ASPP/Decoder, HRNetv2 blocks, LovaszSoftmax,
  TwoScaleLoss, confusion matrix / mIoU, LR schedule, plus the known-answer
  literals listed in SURVEY.md §8c.
* PARITY UNPINNED for the torchvision ResNet arithmetic (``oracle/resnet_tv.py``):
  the reference calls ``torchvision.models.resnet50`` (unpinned version, not
  vendored, not installed in the build image, no network), so that part is a
  restatement of torchvision's published architecture and can only be checked
  for key names / parameter counts / shapes the reference reads.
"""
