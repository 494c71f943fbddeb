#!/usr/bin/env python3
"""Build container only: C++ structs with the FIELDS of the reference's ROS message definitions (cslam_msgs/msg/*.msg -> ccmslam_msgs/<Name>.h), the way
genmsg would lay them out (fixed arrays -> boost::array, variable arrays -> std::vector), written into oracle/ref_shim/ros_stubs/ccmslam_msgs/.
TEST INFRASTRUCTURE: lets `make -C shim check_real` parse the shim translation units against the reference's REAL KeyFrame.h / MapPoint.h / Map.h / Frame.h."""
import os, re, sys
SRC = "/root/reference/cslam_msgs/msg"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim", "ros_stubs", "ccmslam_msgs")
PRIM = {"bool": "uint8_t", "int8": "int8_t", "uint8": "uint8_t", "int16": "int16_t", "uint16": "uint16_t", "int32": "int32_t", "uint32": "uint32_t",
        "int64": "int64_t", "uint64": "uint64_t", "float32": "float", "float64": "double", "string": "std::string", "time": "ros::Time", "duration": "ros::Duration"}
os.makedirs(DST, exist_ok=True)
for fn in sorted(os.listdir(SRC)):
    if not fn.endswith(".msg"):
        continue
    name = fn[:-4]
    fields, deps = [], set()
    for line in open(os.path.join(SRC, fn)):
        line = line.split("#")[0].strip()
        if not line or "=" in line:
            continue
        ty, var = line.split()[:2]
        m = re.match(r"([A-Za-z0-9_/]+)(\[(\d*)\])?$", ty)
        base, arr, n = m.group(1), m.group(2), m.group(3)
        if base == "Header":
            base = "std_msgs/Header"
        if base in PRIM:
            ct = PRIM[base]
        else:
            pkg, _, b = base.rpartition("/")
            pkg = pkg or "ccmslam_msgs"
            deps.add(f"{pkg}/{b}.h")
            ct = f"{pkg}::{b}"
        if arr:
            ct = f"boost::array<{ct}, {n}>" if n else f"std::vector<{ct}>"
        fields.append((ct, var))
    with open(os.path.join(DST, name + ".h"), "w") as f:
        f.write(f"// look-alike of the generated <ccmslam_msgs/{name}.h> (TEST INFRASTRUCTURE, scripts/gen_msg_stubs.py): the fields of cslam_msgs/msg/{fn}\n#pragma once\n")
        f.write("#include <cstdint>\n#include <string>\n#include <vector>\n#include <boost/array.hpp>\n#include <boost/shared_ptr.hpp>\n#include <ros/time.h>\n")
        for d in sorted(deps):
            f.write(f"#include <{d}>\n")
        f.write(f"namespace ccmslam_msgs {{\nstruct {name} {{\n")
        for ct, var in fields:
            f.write(f"  typedef {ct} _{var}_type;\n  {ct} {var};\n")   # genmsg's per-field typedefs (KeyFrame.cpp / MapPoint.cpp name them in Converter template arguments)
        f.write(f"  typedef boost::shared_ptr<{name}> Ptr;\n  typedef boost::shared_ptr<{name} const> ConstPtr;\n}};\n")
        f.write(f"typedef boost::shared_ptr<{name}> {name}Ptr;\ntypedef boost::shared_ptr<{name} const> {name}ConstPtr;\n}}\n")
    print("wrote", name, len(fields), "fields")
