"""Stand-in for cv2: the reference's util.py only needs COLORMAP_HOT at def time."""
COLORMAP_HOT = 11
