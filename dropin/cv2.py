"""Import-time stand-in for OpenCV.

The reference imports cv2 in utils.py:10 / train.py / valid.py but, on the hot path, only ever CALLS it inside
utils.pnp (cv2.solvePnP + cv2.Rodrigues) - which this repo replaces with the HIP PnP kernel.  The image there has no
OpenCV wheel and no network, so the unchanged train.py / valid.py need a module named cv2 to import; any attribute
access raises, so nothing can silently depend on it.
"""


def __getattr__(name):
    raise AttributeError("cv2.%s is not available: the singleshotpose_amd drop-in routes pose recovery through "
                         "singleshotpose_amd.utils.pnp (HIP) and does not ship OpenCV" % name)
