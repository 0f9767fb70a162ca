"""Writes tests/golden/configs_model_sections.json: the parsed `task.dataset` + `model` section of every reference config
(/root/reference/configs/*.yaml, DEFAULT document, the way src/util.py:38-47 `read_yaml` + main.py:23-25 read them).  Data
only (parsed YAML values); run in the build container:   python tests/golden/gen_config_fixture.py
The GPU test test_get_model_constructs_from_every_reference_config builds every system model from it."""
import glob
import json
import os

import yaml

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'configs_model_sections.json')

out = {}
for path in sorted(glob.glob(os.path.join(REF, 'configs', '*.yaml'))):
    name = os.path.splitext(os.path.basename(path))[0]
    with open(path) as f:
        for doc in yaml.safe_load_all(f):
            if doc and doc.get('name') == 'DEFAULT' and 'model' in doc.get('params', {}):
                p = doc['params']
                out[name] = {'task': {'dataset': p['task']['dataset'], 'batch_size': p['task'].get('batch_size')},
                             'model': p['model'], 'random_seed': p.get('random_seed')}
with open(OUT, 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
print('wrote', OUT, sorted(out))
