"""nerfstudio.model_components.losses: the one name model.py:19 imports."""
from torch import nn

MSELoss = nn.MSELoss
L1Loss = nn.L1Loss
