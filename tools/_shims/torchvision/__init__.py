"""Stand-in for torchvision: names only; the golden harness never runs the conv trunk."""
from . import transforms, models  # noqa
