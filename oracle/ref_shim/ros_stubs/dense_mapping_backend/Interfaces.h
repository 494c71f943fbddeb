// look-alike of <dense_mapping_backend/Interfaces.h> (TEST INFRASTRUCTURE): an optional dependency of the reference
#pragma once
