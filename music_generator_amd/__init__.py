"""MI355X-native progressive-GAN training step (drop-in for trainer/models.py + the schedule of
trainer/task.py of yefeza/music-generator).  See DESIGN.md at the repository root."""
from . import backend  # noqa: F401
from .backend import clear_session  # noqa: F401
from .graph import Input, Model  # noqa: F401
from .models import (Adam, GANMonitor, MinibatchStdDev, SoftRectifier, StaticOptTanh, WGAN, WeightedSum,  # noqa: F401
                     add_discriminator_block, add_generator_block, define_composite, define_discriminator,
                     define_generator, discriminator_loss, generator_loss)

__all__ = ["backend", "clear_session", "Input", "Model", "Adam", "GANMonitor", "MinibatchStdDev", "SoftRectifier",
           "StaticOptTanh", "WGAN", "WeightedSum", "add_discriminator_block", "add_generator_block",
           "define_composite", "define_discriminator", "define_generator", "discriminator_loss", "generator_loss"]
