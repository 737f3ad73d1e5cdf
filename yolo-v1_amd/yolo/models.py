"""Model surface of the reference (src/yolo/models.py) with the HIP engine underneath.

Module classes, constructor signatures, attribute names and ``state_dict`` keys are the reference's
(SURVEY.md 8b): the fp32 parameters live in ordinary ``nn.Conv2d`` / ``nn.Linear`` children
(``backbone.features.N``, ``head.1`` / ``head.4``, ``head.conv_layers.N`` / ``head.fc_layers.N``) so
checkpoints written by the reference load unchanged.  What differs is ``forward``:

  * device tensors never reach those children -- a whole conv/pool/FC stack runs as ONE autograd node
    on libyolo_hip.so (engine.Plan): zero-haloed NHWC bf16 activations, MFMA implicit-GEMM convs with
    fused bias + LeakyReLU(0.1), fp32 accumulation;
  * CPU tensors take the stock ``torch.nn`` path -- the reference's own ``--device cpu`` behaviour
    (an explicit device choice, not a fallback: a device tensor without the HIP library raises).
"""

from __future__ import annotations

import copy
import weakref

import torch
import torch.nn as nn

from . import engine


class _PlanOwner:
    """mix-in of the modules that own an engine plan: the plan knows its owner (hooks of yolo.optim.Adam.attach_plan(overlap=True)),
    and a deep copy of the module first waits for a background update of its Linear layers still running on a second stream
    (the copy reads every parameter on the current stream)."""

    def _own(self, plan: "engine.Plan") -> "engine.Plan":
        plan.owner = weakref.ref(self)
        return plan

    def __deepcopy__(self, memo):
        plan = self.__dict__.get("_plan")
        if isinstance(plan, engine.Plan):
            plan.params_ready.wait(keep=True)
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__setstate__(copy.deepcopy(self.__dict__, memo))       # what copy.deepcopy does for an nn.Module without this method
        return new


class Backbone(nn.Module):
    """Abstract feature extractor: subclasses map (N,3,H,W) images to (N,C,H',W') features."""

    def __init__(self):
        super().__init__()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("Subclasses must implement forward method")


def _conv_act(cin: int, cout: int, k: int, stride: int = 1, pad: int = 0) -> list[nn.Module]:
    return [nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad), nn.LeakyReLU(0.1)]


class YOLOv1Backbone(_PlanOwner, Backbone):
    """The 24-convolution network of the YOLOv1 paper: 448x448x3 -> 7x7x1024.

    Layer list = src/yolo/models.py:47-84 of the reference (indices inside ``features`` are part of
    the checkpoint contract)."""

    def __init__(self):
        super().__init__()
        mods: list[nn.Module] = []
        mods += _conv_act(3, 64, 7, 2, 3) + [nn.MaxPool2d(2, 2)]
        mods += _conv_act(64, 192, 3, 1, 1) + [nn.MaxPool2d(2, 2)]
        mods += _conv_act(192, 128, 1) + _conv_act(128, 256, 3, 1, 1) + _conv_act(256, 256, 1) + _conv_act(256, 512, 3, 1, 1) + [nn.MaxPool2d(2, 2)]
        mods += self._make_conv_block(512, 256, 512, 4)
        mods += _conv_act(512, 512, 1) + _conv_act(512, 1024, 3, 1, 1) + [nn.MaxPool2d(2, 2)]
        mods += self._make_conv_block(1024, 512, 1024, 2)
        mods += _conv_act(1024, 1024, 3, 1, 1) + _conv_act(1024, 1024, 3, 2, 1)
        mods += _conv_act(1024, 1024, 3, 1, 1) + _conv_act(1024, 1024, 3, 1, 1)
        self.features = nn.Sequential(*mods)
        self._plan: engine.Plan | None = None

    def _make_conv_block(self, in_channels: int, mid_channels: int, out_channels: int, num_blocks: int) -> list[nn.Module]:
        """``num_blocks`` x [1x1 reduce -> 3x3 expand], each followed by LeakyReLU(0.1)."""
        out: list[nn.Module] = []
        for _ in range(num_blocks):
            out += _conv_act(in_channels, mid_channels, 1) + _conv_act(mid_channels, out_channels, 3, 1, 1)
            in_channels = out_channels
        return out

    def hip_plan(self) -> engine.Plan:
        if self._plan is None:
            self._plan = engine.Plan.from_modules(self.features, 3, True)
        return self._own(self._plan)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda:
            return engine.run_plan(self.hip_plan(), x, self.training)
        return self.features(x)


class ResNetBackbone(Backbone):
    """ResNet50 trunk up to layer4 (2048 x 14 x 14 for 448 x 448 inputs), reference src/yolo/models.py:131-176.

    The reference wraps ``torchvision.models.resnet50``; torchvision is an un-vendored dependency, so the
    trunk is restated in ``yolo.resnet`` with torchvision's module names (``extractor.N...`` state_dict keys
    are the reference's).  ``pretrained=True`` needs the ImageNet weights, i.e. torchvision + a download.
    Device tensors run on the HIP engine: eval mode with BatchNorm folded and the residual add fused, training mode
    (frozen backbone) with batch-statistics BatchNorm; a trainable trunk keeps every unit's conv output and runs the backward pass
    on the device in both modes (batch statistics in train(), running statistics in eval())."""

    def __init__(self, pretrained: bool = True, freeze: bool = True):
        super().__init__()
        from .resnet import resnet50_trunk
        trunk = resnet50_trunk()
        if pretrained:
            try:
                from torchvision.models import ResNet50_Weights, resnet50
            except ImportError as e:
                raise ImportError("ResNetBackbone(pretrained=True) needs torchvision's ImageNet weights; "
                                  "use pretrained=False or load a checkpoint") from e
            tv = nn.Sequential(*list(resnet50(weights=ResNet50_Weights.DEFAULT).children())[:-2])
            trunk.load_state_dict(tv.state_dict())
        if freeze:
            for p in trunk.parameters():
                p.requires_grad = False
        self.extractor = trunk
        self._plan = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda:
            if self._plan is None:
                self._plan = engine.ResNetPlan(self.extractor)
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                # trainable trunk: in training mode -- the reference's default run (src/train.py:144) -- BatchNorm normalises with batch
                # statistics; in eval() mode with the running statistics as they are (stock autograd through batch_norm(training=False))
                return engine.ResNetTrainFunction.apply(self._plan, not self.training, x, *self.extractor.parameters())
            if self.training:
                # frozen but in training mode: BatchNorm uses batch statistics and updates its running statistics, exactly what
                # the reference does (freeze does not put BN in eval; trainer.py:49 calls model.train() on everything)
                return self._plan.forward_batch_stats(x)
            return self._plan.forward(x)
        return self.extractor(x)


class DetectionHead(_PlanOwner, nn.Module):
    """Conv + FC head used on top of ResNet50 features (src/yolo/models.py:279-348):
    4 x (3x3 conv + LeakyReLU), the second with stride 2 (14x14 -> 7x7), then
    Flatten -> Linear(1024*S*S, 4096) -> LeakyReLU -> Dropout(0.5) -> Linear(4096, S*S*(5B+C))."""

    def __init__(self, in_channels: int, num_classes: int = 20, S: int = 7, B: int = 2) -> None:
        super().__init__()
        self.num_classes, self.S, self.B = num_classes, S, B
        self.conv_layers = nn.Sequential(
            *_conv_act(in_channels, 1024, 3, 1, 1), *_conv_act(1024, 1024, 3, 2, 1),
            *_conv_act(1024, 1024, 3, 1, 1), *_conv_act(1024, 1024, 3, 1, 1))
        self.fc_layers = nn.Sequential(
            nn.Flatten(), nn.Linear(1024 * S * S, 4096), nn.LeakyReLU(0.1), nn.Dropout(0.5),
            nn.Linear(4096, S * S * (B * 5 + num_classes)))
        self._plan: engine.Plan | None = None
        self._in_channels = in_channels

    def hip_plan(self) -> engine.Plan:
        if self._plan is None:
            self._plan = engine.Plan.from_modules(list(self.conv_layers) + list(self.fc_layers), self._in_channels, False)
        return self._own(self._plan)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda:
            y = engine.run_plan(self.hip_plan(), x, self.training)
        else:
            y = self.fc_layers(self.conv_layers(x))
        return y.view(-1, self.S, self.S, self.B * 5 + self.num_classes)


class YOLOv1(_PlanOwner, nn.Module):
    """Backbone + detection head -> (N, S, S, 5B + C) raw predictions (no output activation).

    ``YOLOv1()`` = YOLOv1Backbone + Flatten/Linear/LeakyReLU/Dropout/Linear head, exactly the
    reference's default (src/yolo/models.py:198-276).  For that configuration a device forward
    runs backbone and head as a single HIP plan (no NCHW fp32 round trip between them)."""

    def __init__(self, backbone: Backbone | None = None, detection_head: nn.Module | None = None,
                 num_classes: int = 20, S: int = 7, B: int = 2):
        super().__init__()
        self.num_classes, self.S, self.B = num_classes, S, B
        if backbone is None:
            backbone = YOLOv1Backbone()
        self.backbone = backbone
        if detection_head is None:
            if isinstance(backbone, YOLOv1Backbone):
                detection_head = nn.Sequential(
                    nn.Flatten(), nn.Linear(1024 * S * S, 4096), nn.LeakyReLU(0.1), nn.Dropout(0.5),
                    nn.Linear(4096, S * S * (B * 5 + num_classes)))
            elif isinstance(backbone, ResNetBackbone):
                detection_head = DetectionHead(2048, num_classes, S, B)
            else:
                raise ValueError("Must provide detection_head for custom backbone types")
        self.head = detection_head
        self._plan: engine.Plan | None = None

    def _fusable(self) -> bool:
        h = self.head
        return (type(self.backbone) is YOLOv1Backbone and type(h) is nn.Sequential and len(h) == 5
                and isinstance(h[0], nn.Flatten) and isinstance(h[1], nn.Linear) and isinstance(h[2], nn.LeakyReLU)
                and isinstance(h[3], nn.Dropout) and isinstance(h[4], nn.Linear))

    def hip_plan(self) -> engine.Plan:
        if self._plan is None:
            self._plan = engine.Plan.from_modules(list(self.backbone.features) + list(self.head), 3, True)
        return self._own(self._plan)

    @torch.no_grad()
    def forward_uint8(self, images: torch.Tensor, size: tuple[int, int] = (448, 448)) -> torch.Tensor:
        """Inference from decoded uint8 RGB images [N][H][W][3] on the device: Resize(size) + ToTensor + Normalize
        (the reference's transform, inference.py:58-66, bit-exact) are done by yolo_preprocess_u8 straight into the stem's
        input buffer -- extension of the reference surface for serving; equals ``self(transform(images))``."""
        if not (images.is_cuda and self._fusable()):
            raise RuntimeError("forward_uint8 needs device images and the default YOLOv1 backbone + head")
        y, _ = self.hip_plan().forward(images, False, False, u8_size=size)
        return y.view(-1, self.S, self.S, self.B * 5 + self.num_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda and self._fusable():
            y = engine.run_plan(self.hip_plan(), x, self.training)
        else:
            y = self.head(self.backbone(x))
        if y.dim() == 2:
            y = y.view(-1, self.S, self.S, self.B * 5 + self.num_classes)
        return y
