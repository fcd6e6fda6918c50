'use strict';
const { MlsMpmSimulator, native } = require('./simulator');
module.exports = Object.assign({ MlsMpmSimulator, default: MlsMpmSimulator, native },
    require('./vector3'), require('./structured_array'), require('./config'), require('./particles'),
    { modes: require('./modes') }, require('./app'));
