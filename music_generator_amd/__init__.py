"""MI355X-native progressive-GAN training step (drop-in for trainer/models.py + the schedule of
trainer/task.py of yefeza/music-generator): the Keras-like construction API in models.py / layers.py /
graph.py, every numerical operation a call into libmgx.so (hand-written gfx950 kernels, C-ABI in
include/mgx.h) through ops.py."""
from . import backend  # noqa: F401
from .backend import clear_session  # noqa: F401
from .graph import Input, Model  # noqa: F401
from .inference import ChecksumSink, CollapsedGenerator, HostRingSink, draw_latents, generate_clips  # noqa: F401
from .models import (Adam, GANMonitor, MinibatchStdDev, SoftRectifier, StaticOptTanh, WGAN, WeightedSum,  # noqa: F401
                     add_discriminator_block, add_generator_block, define_composite, define_discriminator,
                     define_generator, discriminator_loss, generator_loss)

__all__ = ["backend", "clear_session", "Input", "Model", "Adam", "GANMonitor", "MinibatchStdDev", "SoftRectifier",
           "StaticOptTanh", "WGAN", "WeightedSum", "add_discriminator_block", "add_generator_block",
           "define_composite", "define_discriminator", "define_generator", "discriminator_loss", "generator_loss",
           "generate_clips", "draw_latents", "ChecksumSink", "HostRingSink", "CollapsedGenerator"]
