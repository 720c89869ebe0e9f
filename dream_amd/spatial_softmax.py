"""SoftArgmaxPavlo on the HIP path (mirrors /root/reference/dream/spatial_softmax.py:15-95).

Forward only, like the reference in practice: training on the keypoint head is "Not yet
implemented" there (dream/network.py:343,361-362).  ``beta`` is a Parameter when ``learned_beta``
(state_dict key ``softmax.0.beta``), otherwise a plain tensor that follows the module's device."""
import torch

from . import ops


class SoftArgmaxPavlo(torch.nn.Module):
    def __init__(self, n_keypoints=5, learned_beta=False, initial_beta=25.0):
        super().__init__()
        beta = torch.ones(n_keypoints) * initial_beta
        if learned_beta:
            self.beta = torch.nn.Parameter(beta)
        else:
            self.register_buffer("_beta_const", beta, persistent=False)

    def _beta(self):
        return self.beta if "beta" in self._parameters else self._beta_const

    def forward(self, heatmaps, size_mult=1.0):
        beta = self._beta().detach().to(heatmaps.device)
        with torch.no_grad():
            return ops.softargmax(heatmaps.detach(), beta, size_mult)
