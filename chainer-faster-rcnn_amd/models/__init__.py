"""Host-side mirror of the reference's `models/` package: same module / class / function names
(ProposalLayer, cpu_nms, roi_pooling_2d, VGG16Prev, RegionProposalNetwork, FasterRCNN, ...), each backed by
the HIP kernels in libfrcnn_hip.so."""
from .anchor_target_layer import AnchorTargetLayer  # noqa: F401
from .bbox import bbox_overlaps  # noqa: F401
from .bbox_transform import bbox_transform_inv, clip_boxes  # noqa: F401
from .cpu_nms import cpu_nms  # noqa: F401
from .gpu_nms import gpu_nms  # noqa: F401
from .faster_rcnn import FasterRCNN  # noqa: F401
from .generate_anchors import generate_anchors  # noqa: F401
from .proposal_layer import ProposalLayer  # noqa: F401
from .proposal_target_layer import ProposalTargetLayer  # noqa: F401
from .region_proposal_network import RegionProposalNetwork  # noqa: F401
from .resnet import ResNet, ResNet50, ResNet101, ResNet152  # noqa: F401
from .roi_pooling_2d import ROIPooling2D, roi_pooling_2d  # noqa: F401
from .vgg16 import VGG16, VGG16Prev  # noqa: F401
