from .nerf import NeRFRenderer
