"""Regenerates oracle/orb_pattern.h and ccm_slam_amd/csrc/orb_pattern.h from the reference's
bit_pattern_31_ table (cslam/src/ORBextractor.cpp:319-577).  Runs only where /root/reference exists;
the generated headers are committed."""
import re, sys
src = open('/root/reference/cslam/src/ORBextractor.cpp').read()
a = src.index('static int bit_pattern_31_[256*4] =')
body = src[a:src.index('};', a)]
body = re.sub(r'/\*.*?\*/', '', body[body.index('{') + 1:], flags=re.S)
nums = [int(x) for x in re.findall(r'-?\d+', body)]
assert len(nums) == 1024
print(len(nums), "values; first pair:", nums[:4])
