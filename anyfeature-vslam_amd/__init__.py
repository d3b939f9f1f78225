"""anyfeature-vslam_amd — MI355X-native ORB32 extraction + descriptor matching behind AnyFeature-VSLAM's
FeatureExtractor / FeatureMatcher interface.  (The directory name carries a hyphen: import it with
importlib.import_module("anyfeature-vslam_amd").)

Layout:
  csrc/          hand-written HIP kernels for gfx950 + the C-ABI runtime (libafv_hip.so, include/afv_hip.h)
  adapter/       C++ host adapter mirroring the reference's classes on top of the C-ABI
  _lib.py        ctypes binding
  extractor.py   FeatureExtractorSettings / FeatureExtractor_orb32 mirror (reference: Feature_orb32.{h,cpp})
  matcher.py     FeatureMatcher mirror (reference: FeatureMatcher.{h,cc})
  frame.py       the device-resident Frame (reference: Frame.{h,cc} as far as the front end reads it)
  synth.py       bit-reproducible synthetic inputs
"""
from . import _lib, synth  # noqa: F401
from .extractor import (CovarianceMethod, FeatureExtractorSettings, FeatureExtractor_orb32, Context,  # noqa: F401
                        KP_DTYPE)
from .vocabulary import Vocabulary  # noqa: F401
from .frame import Frame  # noqa: F401
from . import akaze, table  # noqa: F401
from .akaze import AkazeContext  # noqa: F401
from .matcher import (FeatureMatcher, FeatureView, FrameGridView, ProjectionQueries, ComputeDistinctiveDescriptors,  # noqa: F401
                      DescriptorDistance_orb32, DescriptorDistance_sift128)
