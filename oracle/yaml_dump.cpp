// ORACLE (test infrastructure, NOT product code): dumps what the reference's own yaml dependency (yaml-cpp 0.6.2, vendored
// under 3rdPartLib/, linked from its sources by oracle/Makefile into oracle/_ref/) reads from a config file, using the
// same accessors as src/utils/include/yamlRead.h:7-66 (config[key].as<double>() / .as<std::vector<double>>()).
// Output: one line per top-level key, `key n v0 v1 ...` with %.17g values; keys whose value is not numeric are printed as
// `key s <text>`.  Used to pin flvis_config_load and the oracle's loader on the reference's launch/*.yaml files
// (tests/golden/yaml_*.txt, scripts/make_yaml_fixtures.py).
#include <cstdio>
#include <string>
#include <vector>

#include "yaml-cpp/yaml.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  YAML::Node config = YAML::LoadFile(argv[1]);
  for (YAML::const_iterator it = config.begin(); it != config.end(); ++it) {
    const std::string key = it->first.as<std::string>();
    const YAML::Node& v = it->second;
    try {
      if (v.IsSequence()) {
        const std::vector<double> d = v.as<std::vector<double>>();
        std::printf("%s %zu", key.c_str(), d.size());
        for (double x : d) std::printf(" %.17g", x);
        std::printf("\n");
      } else {
        const double d = v.as<double>();
        std::printf("%s 1 %.17g\n", key.c_str(), d);
      }
    } catch (const YAML::Exception&) {
      std::printf("%s s %s\n", key.c_str(), v.IsScalar() ? v.as<std::string>().c_str() : "<non-scalar>");
    }
  }
  return 0;
}
