"""nerfstudio.utils.colors (0.3.4)."""
import torch

WHITE = torch.tensor([1.0, 1.0, 1.0])
BLACK = torch.tensor([0.0, 0.0, 0.0])
RED = torch.tensor([1.0, 0.0, 0.0])
GREEN = torch.tensor([0.0, 1.0, 0.0])
BLUE = torch.tensor([0.0, 0.0, 1.0])

COLORS_DICT = {"white": WHITE, "black": BLACK, "red": RED, "green": GREEN, "blue": BLUE}
