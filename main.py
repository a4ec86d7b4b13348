"""ID-GRec entry point on MI355X.

    python main.py --model=LightGCN

Same four steps, console output, log file layout (log/<Model>/<dataset>.log) and plugin
protocol (`from models.<Model> import Trainer`; `Trainer(args, config, dataset, device,
logger).train()`) as the reference's main.py; the work underneath runs in libidgrec.so.
"""
import importlib
import logging
import os

import torch

import Parser
import utility.utility_data.data_loader as data_loader
import utility.utility_function.tools as tools

RULE = '-' * 100
# menu numbering of the reference (main.py:37-42); entries without a models/<Name>.py here are
# outside the LightGCN hot path this repository implements
MODEL_MENU = ["MFBPR", "GCMC", "GCCF", "NGCF", "LightGCN", "IMPGCN", "SGL", "CVGA", "SimGCL", "XSimGCL", "DirectAU",
              "NCL", "HCCF", "LightGCL", "DCCF", "CGCL", "MAWU", "RecDCL", "BIGCF", "SCCF", "EGCF", "LightGODE",
              "LightGCN_pp", "MixRec", "LightCCF", "LightCSCF"]


def show_menu():
    cells = ["%d.%s" % (i + 1, name) for i, name in enumerate(MODEL_MENU)]
    for row in range(0, len(cells), 5):
        print('\t ' + ' \t '.join(c.ljust(12) for c in cells[row:row + 5]))


def choose_model(args):
    if args.model != "unknown":
        return args.model
    while True:
        picked = input('Please input the identifier of the model:')
        if picked.isdigit() and 1 <= int(picked) <= len(MODEL_MENU):
            return MODEL_MENU[int(picked) - 1]
        print("Input Error. Please select from the list of implemented models and try again.")


def open_logger(model_name, dataset_name):
    folder = os.path.join('log', model_name)
    if not os.path.exists(folder):
        os.makedirs(folder)
    logger = logging.getLogger('logger')
    logger.setLevel(logging.INFO)
    handler = logging.FileHandler('log/{}/{}.log'.format(model_name, dataset_name), 'a', encoding='utf-8')
    handler.setLevel(logging.INFO)
    handler.setFormatter(logging.Formatter('%(asctime)s - %(message)s'))
    logger.addHandler(handler)
    return logger


def main():
    print('ID-GRec: PyTorch Implementation of ID-based Graph Recommender Systems')
    print(RULE)
    print('Step 1: General parameter setting reading...')
    print(RULE)
    args = Parser.parse_args()
    if args.cuda:
        os.environ["CUDA_VISIBLE_DEVICES"] = str(args.gpu_id)
    device = torch.device('cuda' if torch.cuda.is_available() else "cpu")
    if args.seed_flag:
        tools.set_seed(args.seed)

    print('Step 2: Select model...')
    show_menu()
    print(RULE)
    model_name = choose_model(args)

    print('Step 3.1: Loading configuration file...')
    try:
        trainer_cls = importlib.import_module('models.' + model_name).Trainer
    except ModuleNotFoundError as err:
        raise SystemExit("models/%s.py is not part of this build (%s). Implemented: MFBPR, LightGCN, SimGCL, XSimGCL, SGL, NGCF, EGCF."
                         % (model_name, err))
    config = tools.read_configuration('./configure/' + model_name + ".txt", model_name)
    logger = open_logger(model_name, config['dataset'])

    print('Step 3.2: Loading dataset file...')
    dataset = data_loader.Data(config['dataset_path'] + config['dataset'], config)
    logger.info("Run with " + model_name + " on " + config['dataset'])
    logger.info(dataset.get_statistics())

    print(RULE)
    print('\t Step 3.3: Init the Recommendation Model:')
    recommender = trainer_cls(args, config, dataset, device, logger)
    print('\t model: ', model_name)
    for key in config:
        print("\t " + str(key) + " : " + str(config[key]))
        logger.info(str(key) + " : " + str(config[key]))

    print(RULE)
    print("Step 4: Model training and testing process:")
    recommender.train()


if __name__ == '__main__':
    main()
