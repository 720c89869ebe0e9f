"""dream_amd -- MI355X-native implementation of DREAM's keypoint belief-map hot path.

The names below are the ones the reference's scripts use through ``import dream`` for this path
(/root/reference/dream/__init__.py re-exports network, models, image_proc, spatial_softmax);
``import dream_amd as dream`` is the integration recipe."""
__version__ = "0.1.0"

from . import image_proc, models, network, spatial_softmax  # noqa: F401
from .network import (KNOWN_ARCHITECTURES, KNOWN_OPTIMIZERS, DreamNetwork,  # noqa: F401
                      create_network_from_config_data, create_network_from_config_file)
from .models import DreamHourglass, ResnetSimple  # noqa: F401
from .spatial_softmax import SoftArgmaxPavlo  # noqa: F401
from .image_proc import peaks_from_belief_maps  # noqa: F401
from .configs import default_network_config, arch_config_path, manip_config_path  # noqa: F401
